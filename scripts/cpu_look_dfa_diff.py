"""Differential run: the restated look-aware lazy DFA of the oracle (a fresh cache per input) against Python re.
Test infrastructure (oracle only); DESIGN.md section 7 quotes its output."""
import sys, re, random
sys.path.insert(0,'/root/repo')
from oracle import oracle as o
pats=[r'\b(DEBUG|INFO|WARN|ERROR)\b', r'\b\w+@\w+\.com\b', r'\b[a-z]+ing\b[ ,.]', r'\bfoo[0-9]+bar\b|\bbaz[a-z]+qux\b', r'\b\d+\.\d+\b', r'(?m)^[a-z]+: \d+$',
      r'\b(GET|POST|PUT|DELETE|PATCH) /[a-z/]+', r'\b[A-Z][a-z]+ [A-Z][a-z]+\b', r'\b\w+\s+\w+\s+\w+\b', r'[a-z]+\b[ ]+\b[a-z]+\b[ ]+[0-9]+', r'\b\d{3}-\d{4}\b',
      r'\berror\b.{0}[a-z ]+\btimeout\b', r'(?m)^(ERROR|WARN) [a-z]+ [0-9]+$', r'\b[a-z]+_[a-z]+_[a-z]+\b', r'\d+\b\.\b\d+\b\.\d+', r'\B[a-z]+\B[0-9][0-9][0-9]']
rng=random.Random(5)
alpha="ab fo_9.\nERINFO:@cmg-"
for p in pats:
    try: rx=o.Regex(p)
    except Exception as e: print(p,'ERR',e); continue
    strat=rx.strategy
    bad=0; tot=0; ex=None
    pr=re.compile(p.encode())
    for t in range(300):
        n=rng.randint(0,40)
        words=["foo","INFO","ERROR"," ","_","a","9",".","\n","ing","bar","baz","qux","GET /a/b","x@y.com","12","err: 5","Ab Cd","error","timeout","123-4567","a_b_c"]
        h="".join(rng.choice(words) if rng.random()<0.5 else rng.choice(alpha) for _ in range(n)).encode()
        rx=o.Regex(p)  # fresh cache
        got=[tuple(r) for r in rx.find_all_index(h).tolist()]
        want=[m.span() for m in pr.finditer(h)]
        tot+=1
        if got!=want:
            bad+=1
            if ex is None: ex=(h,got[:3],want[:3])
    print(f"{p:50s} {strat:18s} nfa={rx.nfa_states:3d} alpha={rx.alphabet_len:2d} mismatch {bad}/{tot}", ex if ex else '')
