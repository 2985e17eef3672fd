"""The class-plan arithmetic of coregex_amd/csrc/device/wave_common.hpp (plan_class / notplan4, round 5) restated in Python and checked
exhaustively: for unions of up to four ASCII ranges the X terms ((t ^ lo) + (0x7F - span)), R terms and the upper/lower-case fold flag
exactly the members, for all 256 byte values packed four to a dword with every neighbour combination that could carry."""
import itertools
import random

M32 = 0xFFFFFFFF


def plan_class(ranges):
    p = {"x": [], "r": [], "fold": False, "ok": True}
    rr = []
    for a, b in ranges[:4]:
        if b > 0x7F or a > b:
            p["ok"] = False
            continue
        span, bits = b - a, 0
        while (1 << bits) <= span:
            bits += 1
        if a & ((1 << bits) - 1) == 0:
            p["x"].append((a * 0x01010101, (0x7F - span) * 0x01010101))
        else:
            rr.append((a, b))
    for i, j in itertools.permutations(range(len(rr)), 2):
        if not p["fold"] and rr[i][0] >= 0x40 and rr[i][1] <= 0x5F and rr[j] == (rr[i][0] + 0x20, rr[i][1] + 0x20):
            p["fold"] = True
            rr = [rr[i]] + [rr[q] for q in range(len(rr)) if q not in (i, j)]
            break
    p["r"] = [((0x80 - a) * 0x01010101, (0x7F - b) * 0x01010101) for a, b in rr]
    return p


def notplan4(x, p):
    t = x & 0x7F7F7F7F
    inn = 0
    for c, k in p["x"]:
        inn |= ~(((t ^ c) + k) & M32) & M32
    for i, (ra, rb) in enumerate(p["r"]):
        s = (x & 0x5F5F5F5F) if (p["fold"] and i == 0) else t
        inn |= ((s + ra) & M32) & ~((s + rb) & M32) & M32
    return ~(inn & ~x) & 0x80808080 & M32


CLASSES = {
    r"\w": [(0x30, 0x39), (0x41, 0x5A), (0x5F, 0x5F), (0x61, 0x7A)],
    r"\d": [(0x30, 0x39)],
    "[a-z]": [(0x61, 0x7A)],
    "[A-Za-z]": [(0x41, 0x5A), (0x61, 0x7A)],
    "[A-Fa-f0-9]": [(0x30, 0x39), (0x41, 0x46), (0x61, 0x66)],
    "[ -~]": [(0x20, 0x7E)],
    "[\\x00-\\x1f]": [(0x00, 0x1F)],
    "[@-_`-\\x7f]": [(0x40, 0x5F), (0x60, 0x7F)],
    "[.,;]": [(0x2C, 0x2C), (0x2E, 0x2E), (0x3B, 0x3B)],
    "[!-/:-@]": [(0x21, 0x2F), (0x3A, 0x40)],
    "[B-Yb-y5]": [(0x35, 0x35), (0x42, 0x59), (0x62, 0x79)],
}


def test_plans_flag_exactly_the_members():
    rng = random.Random(7)
    for name, ranges in CLASSES.items():
        p = plan_class(ranges)
        assert p["ok"], name
        member = [any(a <= b <= c for a, c in ranges) for b in range(256)]
        for b in range(256):
            for _ in range(6):                                     # the byte in every position, random neighbours (carries must not cross bytes)
                pos = rng.randrange(4)
                bs = [rng.choice([0x00, 0x7F, 0x80, 0xFF, rng.randrange(256)]) for _ in range(4)]
                bs[pos] = b
                x = bs[0] | bs[1] << 8 | bs[2] << 16 | bs[3] << 24
                got = notplan4(x, p)
                for q in range(4):
                    assert ((got >> (8 * q + 7)) & 1) == (0 if member[bs[q]] else 1), (name, [hex(v) for v in bs], q)
                    assert (got >> (8 * q)) & 0x7F == 0


def test_word_class_is_two_x_terms_and_one_folded_range():
    p = plan_class(CLASSES[r"\w"])
    assert len(p["x"]) == 2 and len(p["r"]) == 1 and p["fold"]
    assert not plan_class([(0x30, 0x39), (0x80, 0x90)])["ok"]
