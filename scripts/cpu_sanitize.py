"""Feeds tests/emu/san_driver (g++ -fsanitize=address,undefined build of host/frontend.cc, host/program.cc, host/fsm.cc and the
transducer twin) with the patterns of the test suite plus random ones.  python scripts/cpu_sanitize.py [n_random] [seed]"""
import sys, os, subprocess, json, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import cpu_fuzz_fsm as F
import cpu_fuzz_lookdfa as L

def main(n=3000, seed=1):
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "emu"), "san"])
    rng = random.Random(seed)
    pats = set()
    vec = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_vectors.json")))
    def walk(x):
        if isinstance(x, dict):
            for k, v in x.items():
                if k == "pattern" and isinstance(v, str): pats.add(v)
                else: walk(v)
        elif isinstance(x, list):
            for v in x: walk(v)
    walk(vec)
    atoms = F.ATOMS + F.LOOK_ATOMS + L.ATOMS * 2 + F.WIDE_ATOMS * 2 + F.FOLD_ATOMS * 2 + ["[\\x{100}-\\x{7FF}]", "[\\x{800}-\\x{FFFF}]", "[^\\x00-\\x7F]", "(?i:exception)", "(?i:kkkkkk)", "^", "$", "\\z", "\\A", "(?i)(error|fail|panic)", "(?i:(?:jan|jun|jul))", "(?:^|,)", "(^|\\s)", "(?:a*)*"] + ["(", ")", "[", "]", "{2,", "}", "|", "*", "+", "?", "\\", "(?i)", "(?m)", "(?s)", ".", "[^a]", r"\x41", r"\pL", "{1000}", "(?:", "(?P<n>a)", "[a-", "\\Q.\\E", "a{,3}"]
    while len(pats) < n:
        pats.add("".join(rng.choice(atoms) for _ in range(rng.randint(1, 6))))
    data = "\n".join(p for p in pats if "\n" not in p).encode()
    r = subprocess.run([os.path.join(ROOT, "tests", "emu", "san_driver")], input=data, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=3000)
    sys.stdout.write(r.stdout.decode())
    if r.returncode != 0:
        sys.stdout.write(r.stderr.decode()[-6000:])
    return r.returncode

if __name__ == "__main__":
    sys.exit(main(int(sys.argv[1]) if len(sys.argv) > 1 else 3000, int(sys.argv[2]) if len(sys.argv) > 2 else 1))
