"""Prototype of the FindAll transducer (stack of searches) + uncertainty automaton; sizes and a differential check."""
import sys, itertools, time
sys.path.insert(0, '/root/repo')
import numpy as np
import coregex_amd as cx

MATCH, BR, SPARSE, SPLIT, EPS, CAP, FAIL, LOOK = range(8)
INV = 0xFFFFFFFF

class N:
    def __init__(s, v):
        s.states = [(st.kind, st.lo, st.hi, st.next, st.left, st.right, st.trans_off, st.trans_len) for st in (v.states[i] for i in range(v.n_states))]
        s.trans = [(t.lo, t.hi, t.next) for t in (v.trans[i] for i in range(v.n_trans))]
        s.sa, s.su = v.start_anchored, v.start_unanchored

def closure_into(n, out, seen, seed):
    stack = [seed]
    while stack:
        cur = stack.pop()
        if cur == INV or cur >= len(n.states) or cur in seen: continue
        seen.add(cur); out.append(cur)
        k = n.states[cur]
        if k[0] in (EPS, CAP): stack.append(k[3])
        elif k[0] == SPLIT: stack.append(k[5]); stack.append(k[4])

def step(n, lst, b):
    out, seen = [], set()
    for sid in lst:
        k = n.states[sid]
        if k[0] == BR:
            if k[1] <= b <= k[2]: closure_into(n, out, seen, k[3])
        elif k[0] == SPARSE:
            for t in n.trans[k[6]:k[6]+k[7]]:
                if t[0] <= b <= t[1]: closure_into(n, out, seen, t[2])
    return out

def match_index(n, lst):
    for i, s in enumerate(lst):
        if n.states[s][0] == MATCH: return i
    return -1

def reps_of(n):
    bnd = [False]*256
    def mark(lo, hi):
        if lo > 0: bnd[lo-1] = True
        bnd[hi] = True
    for k in n.states:
        if k[0] == BR: mark(k[1], k[2])
        elif k[0] == SPARSE:
            for t in n.trans[k[6]:k[6]+k[7]]: mark(t[0], t[1])
    reps, cls, c = [], [0]*256, 0
    for b in range(256):
        if b == 0 or bnd[b-1]: reps.append(b)
        cls[b] = len(reps)-1
    return reps, cls

def build(n, max_states=4000, max_depth=6):
    reps, cls = reps_of(n)
    fresh = []; closure_into(n, fresh, set(), n.su)
    fresh = tuple(fresh)
    assert match_index(n, fresh) < 0, "nullable"
    ids = {}; states = []; trans = []; events = []
    def intern(x):
        if x in ids: return ids[x]
        if len(x) > max_depth: raise RuntimeError("depth")
        if len(states) >= max_states: raise RuntimeError("states")
        ids[x] = len(states); states.append(x); trans.append(None); events.append(None)
        return ids[x]
    intern((fresh,))
    cur = 0
    while cur < len(states):
        X = states[cur]
        row, ev = [], []
        for rb in reps:
            lv = [step(n, list(L), rb) for L in X]
            k = len(X) - 1
            new = []; e = None; died = 0
            done = False
            for j in range(k):
                m = match_index(n, lv[j])
                if m >= 0:
                    conts = tuple(lv[j][:m])
                    if conts: new.append(conts)
                    new.append(fresh)
                    e = ('rematch', j, bool(conts), died)
                    done = True
                    break
                if not lv[j]: died |= 1 << j
                else: new.append(tuple(lv[j]))
            if not done:
                m = match_index(n, lv[k])
                if m >= 0:
                    conts = tuple(lv[k][:m])
                    if conts: new.append(conts)
                    new.append(fresh)
                    e = ('create', k, bool(conts), died)
                else:
                    new.append(tuple(lv[k]))
                    e = ('none', 0, False, died) if died else None
            row.append(intern(tuple(new))); ev.append(e)
        trans[cur] = row; events[cur] = ev
        cur += 1
    return dict(reps=reps, cls=cls, states=states, trans=trans, events=events)

def uncertainty(T, cap=2000):
    ns = len(T['states']); nc = len(T['reps'])
    top = frozenset(range(ns))
    ids = {top: 0}; sets = [top]; tr = []
    cur = 0
    while cur < len(sets):
        S = sets[cur]; row = []
        for c in range(nc):
            img = frozenset(T['trans'][s][c] for s in S)
            if len(img) == 1: row.append(('s', next(iter(img))))
            else:
                if img not in ids:
                    if len(sets) >= cap: raise RuntimeError("U cap")
                    ids[img] = len(sets); sets.append(img)
                row.append(('u', ids[img]))
        tr.append(row); cur += 1
    return sets, tr

def findall_T(T, hay):
    """sequential replay with row bookkeeping: returns committed ends list"""
    x = 0; rows = []; stack = []  # stack of row indices for live cont levels
    cls = T['cls']
    for i, b in enumerate(hay):
        e = T['events'][x][cls[b]]
        x = T['trans'][x][cls[b]]
        if e is None: continue
        kind, j, conts, died = e
        if kind == 'rematch':
            st2 = [stack[q] for q in range(j) if not (died >> q) & 1]
            r = stack[j]
            del rows[r+1:]
            rows[r] = i + 1
            if conts: st2.append(r)
            stack = st2
        else:
            stack = [stack[q] for q in range(len(stack)) if not (died >> q) & 1]
            if kind == 'create':
                rows.append(i + 1)
                if conts: stack.append(len(rows) - 1)
    return rows

if __name__ == '__main__':
    from oracle import oracle as O
    pats = [r"\d+\.\d+\.\d+\.\d+", r"error", r"(\w+)@(\w+)\.(\w+)", r"\d+\.\d+x?", r"ab|abc", r"a[0-9]*b|a\.", r"a+b|b+a",
            r"(?:25[0-5]|2[0-4][0-9]|[01]?[0-9][0-9]?)\.(?:25[0-5]|2[0-4][0-9]|[01]?[0-9][0-9]?)\.(?:25[0-5]|2[0-4][0-9]|[01]?[0-9][0-9]?)\.(?:25[0-5]|2[0-4][0-9]|[01]?[0-9][0-9]?)",
            r"(?:0[0-9]|1[0-9]|2[0-3]):[0-5][0-9]:[0-5][0-9]", r"ab*c|a|bb", r"(foobar|foo)\d*", r"[a-z]+=\d+", r"HTTP/\d\.\d", r"x[ab]+?y", r"a(b*c)?", r"\d{1,3}\.\d{1,3}\.\d{1,3}\.\d{1,3}",
            r"[a-f0-9]{8}-[a-f0-9]{4}", r"(GET|POST|PUT) /[a-z/]+", r"\w+\.txt", r"[1-9][0-9]*|0", r"\d+[a-z]", r"(?:ab)+c", r"a{2,4}b"]
    rng = np.random.default_rng(1)
    for pat in pats:
        rx = cx.compile(pat)
        n = N(rx.nfa())
        t0 = time.time()
        try:
            T = build(n)
        except RuntimeError as ex:
            print(pat, 'FAIL', ex); continue
        depth = max(len(x) for x in T['states'])
        try:
            sets, utr = uncertainty(T)
            nu = len(sets); maxsz = max(len(s) for s in sets[1:]) if nu > 1 else 0
        except RuntimeError as ex:
            nu = -1; maxsz = -1
        print(f"{pat[:50]:50s} {rx.strategy:18s} nfa={len(n.states):3d} cls={len(T['reps']):2d} T={len(T['states']):4d} depth={depth} U={nu} maxset={maxsz} {time.time()-t0:.2f}s")
        o = O.Regex(pat)
        alph = np.frombuffer(b"abcfoxy.:-0123456789 =/GETPOS\n@_", dtype=np.uint8)
        for trial in range(30):
            hay = alph[rng.integers(0, len(alph), size=int(rng.integers(1, 400)))]
            if trial % 3 == 0: hay = alph[rng.integers(0, 6, size=300)]
            exp = o.find_all_submatch_index(hay)[:, 1].tolist() if o.strategy in ('UseBoth',) else o.find_all_index(hay)[:, 1].tolist()
            got = findall_T(T, hay.tobytes())
            if got != exp:
                print('  MISMATCH', pat, hay.tobytes()[:80], got[:10], exp[:10]); break
