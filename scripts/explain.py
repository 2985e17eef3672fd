#!/usr/bin/env python3
"""What would the library do with a pattern?  Strategy (the reference's, as the front-end derives it), whether the device subset serves
it and — if so — the first kernel a call launches, following capi_ladder.hip scanDeviceOnce (fallbacks then follow the ladder of DESIGN §1:
wave kernel -> transducer -> table-walking kernel).  No GPU needed.

  python scripts/explain.py 'PATTERN' ['PATTERN' ...]
  python scripts/explain.py --survey            # the list of everyday log / text patterns below, one line each
"""
import os, struct, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import coregex_amd as cx
import emu                                   # trio_shape / fields_shape of a chain program (the kernels' own predicates)

KIND = {1: "digit", 2: "DFA pair", 3: "char class", 4: "literal set", 5: "transducer only"}


def first_kernel(rx, sub=False):
    blob = rx.submatch_blobs()[0] if sub else rx.blob()
    kind, flags = struct.unpack_from("<II", blob, 4)
    aux_len = struct.unpack_from("<I", blob, 60)[0]
    fsm = rx.fsm_image(sub) is not None
    if kind == 4:
        return "k_scan_teddy_pair (k_scan_teddy_wave behind it: FindAll with an n, match-dense input)" if aux_len <= 2048 else "k_scan_teddy"
    if kind == 3:
        return "k_scan_charclass_wave" if flags & 64 else "k_scan_charclass"
    def fsm_name():                                             # round 6: shallow machines start on the lean kernel (FsmHeader: depth, nk, direct_off)
        img = rx.fsm_image(sub)
        depth, nk, direct = struct.unpack_from("<I", img, 7 * 4)[0], struct.unpack_from("<I", img, 23 * 4)[0], struct.unpack_from("<I", img, 30 * 4)[0]
        if depth > 1:
            return "k_scan_fsm"
        return "k_scan_fsml<direct>" if direct and nk == 1 else "k_scan_fsml (k_scan_fsm for input whose entry states do not collapse)"
    if kind == 5:
        return fsm_name() + " (look-around / large NFA: no other kernel)"
    if flags & 16:                                              # complete ordered chain
        if sub:
            caps = rx.chain_captures() is not None
            shape = emu.trio_shape(blob)
            if shape and caps and not flags & 512:
                return "k_scan_trio_wave<%d%s> (capture rows)" % (shape & 7, ", EQ" if shape & 8 else "")
            return "k_scan_chain_wave<CAP>" if caps else "k_scan_chain_wave, then the capture pass"
        if not flags & (512 | 32) and emu.fields_shape(blob):
            return "k_scan_fields_wave<%d>" % emu.fields_shape(blob)
        shape = emu.trio_shape(blob)
        if shape and not shape & 8 and not flags & 512:
            return "k_scan_trio_wave<%d>" % (shape & 7)
        return "k_scan_chain_wave" + ("<BND>" if flags & 512 else "")
    if kind == 2 and flags & 256:
        return "k_scan_teddy_wave<VERIFY> (literal prefixes + anchored DFA walk)"
    if fsm:
        return fsm_name()
    return "k_scan_digit_flat" if kind == 1 else "k_scan_dfa<bidir>"


def explain(pat):
    try:
        rx = cx.compile(pat)
    except cx.CoregexError as e:
        return {"pattern": pat, "strategy": "-", "served": False, "why": str(e)}
    out = {"pattern": pat, "strategy": rx.strategy, "served": rx.supported, "why": rx.why_unsupported if not rx.supported else "", "groups": rx.num_groups - 1,
           "nfa_states": rx.nfa_states}
    if rx.supported:
        out["kernel"] = first_kernel(rx)
        img = rx.fsm_image()
        out["transducer_image"] = len(img) if img is not None else 0
    if rx.num_groups > 1:
        out["captures"] = rx.submatch_supported
        if rx.submatch_supported and rx.offset_captures is not None:     # round 4: slots at fixed distances from the span's ends
            out["capture_path"] = first_kernel(rx) + " / offsets of the span (k_caps_offsets)"
        elif rx.submatch_supported:
            cb = rx.submatch_blobs()[1]
            caps = "in the span kernel" if rx.chain_captures() is not None else "backtracking pass per row" if cb[:4] == b"TBXC" else "one-pass table"
            out["capture_path"] = first_kernel(rx, True) + " / " + caps
    return out


SURVEY = [
    # the BASELINE configurations
    r"error", r"\d+\.\d+\.\d+\.\d+", r"error|warning|fatal|critical|panic|timeout|refused|denied", r"[\w]+", r"(\w+)@(\w+)\.(\w+)",
    # access / application logs
    r"(\d+)\.(\d+)\.(\d+)\.(\d+)", r'"(GET|POST|PUT|DELETE) ([^ ]+) HTTP/\d\.\d"', r'"[^"]*"', r"\[[^\]]+\]", r"GET .* HTTP", r"HTTP/\d\.\d\" \d{3}", r" [45]\d\d ",
    r"\d{4}-\d{2}-\d{2}", r"\d{2}:\d{2}:\d{2}", r"\d{4}-\d{2}-\d{2}[T ]\d{2}:\d{2}:\d{2}", r"(?m)^\d+\.\d+\.\d+\.\d+", r"(?m)^(GET|POST|PUT|DELETE|PATCH)", r"(?m)^ERROR.*$", r"(?m)^.*error.*$",
    r"user=(\w+)", r"user=(\S+)", r"(\w+)=(\w+)", r"(\w+)=([^ ]+)", r"session_id=[0-9a-f]+", r"[0-9a-f]{32}", r"[0-9a-f]{8}-[0-9a-f]{4}-[0-9a-f]{4}-[0-9a-f]{4}-[0-9a-f]{12}",
    r"(?i)error", r"(?i)(error|fail|exception|panic|fatal)", r"(?i)(?:login|auth|sign.?in|session)", r"(?i)warn(ing)?", r"\bERROR\b", r"\b\d+\b", r"\b(GET|POST)\b",
    r"https?://[^\s]+", r"[\w.+-]+@[\w-]+\.[\w.-]+", r"\S+@\S+", r"<[^>]+>", r"/\*.*?\*/", r"\w+\.(txt|log|md)", r".*\.php", r"[^,]+,", r"\S+",
    r"^\d+", r"foo$", r"a*", r".*", r"\p{L}+", r"(?i)é", r"(\w+)\s+\1",
]


if __name__ == "__main__":
    args = sys.argv[1:]
    if args == ["--survey"]:
        served = 0
        for pat in SURVEY:
            e = explain(pat)
            served += bool(e["served"])
            tail = e.get("kernel", "") if e["served"] else "REFUSED: " + e["why"][:110]
            cap = ""
            if e.get("groups"):
                cap = "   | captures: " + (e.get("capture_path", "") if e.get("captures") else "refused")
            print(f"{pat:62s} {e['strategy']:26s} {tail}{cap}")
        print(f"{served} of {len(SURVEY)} served")
    else:
        for pat in args:
            for k, v in explain(pat).items():
                print(f"{k:18s} {v}")
            print()
