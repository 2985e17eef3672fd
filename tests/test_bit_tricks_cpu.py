"""Round-5 device rewrites restated for the host and checked against the forms they replaced (tests/emu/bit_tricks.cc):
trio_rows' one-word search for a row's link bytes, and the char-class kernel's four-stream staging of starts and ends."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))


def test_one_word_rows_and_four_stream_staging_equal_the_plain_forms(tmp_path):
    exe = tmp_path / "bit_tricks"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", str(exe), os.path.join(HERE, "emu", "bit_tricks.cc")])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "one word vs two words: 0 differ" in out.stdout
    assert "four streams vs plain: 0 differ" in out.stdout
