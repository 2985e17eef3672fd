//go:build hip && cgo

// Batch FindAll path of meta.Engine on an MI355X (libcoregex_hip.so, include/coregex_hip.h).
//
// Drop this file and findall_nohip.go into coregex's meta/ package, the two accessor files into nfa/ and prefilter/, and apply
// hooks.patch (four call sites).  Build with `-tags hip`; without the tag the Engine is byte-for-byte what it is today.
//
// The library has no CPU search path: every failure here (no GPU, unsupported program, CXG_E_INPUT for one haystack)
// returns ok == false and the caller runs the existing Go loop — the reference's own degrade-don't-fail rule
// (meta/find.go:285-303 falls back from the DFA to the PikeVM the same way).
package meta

/*
#cgo CFLAGS: -I${SRCDIR}/../third_party/coregex_hip/include
#cgo LDFLAGS: -L${SRCDIR}/../third_party/coregex_hip -lcoregex_hip_rocm -Wl,-rpath,/opt/rocm/lib
#include <stdlib.h>
#include <string.h>
#include "coregex_hip.h"
*/
import "C"

import (
	"runtime"
	"sync"
	"sync/atomic"
	"time"
	"unsafe"

	"github.com/coregx/coregex/nfa"
	"github.com/coregx/coregex/prefilter"
)

// hipThreshold: below this many bytes the CPU loop wins.  Measured on MI355X (scripts/time_host_path.py of the library's
// repository): a call costs ~50 us + len / (25..55 GB/s of PCIe staging) — 64 KiB in 0.055 ms, 1 MiB in 0.13 ms (8 GB/s: the
// reference's fastest published path, digit-lead on EPYC), 4 MiB in 0.21 ms.
const hipThreshold = 1 << 20

// hipProgram is immutable after buildHipProgram and shared by all goroutines (cxg_program is immutable; every call
// of the library uses per-OS-thread stream + scratch: the SearchState analogue, meta/search_state.go:23-62).
type hipProgram struct {
	p   *C.cxg_program // FindAllIndex / Count program, nil: not served
	sub *C.cxg_program // span + capture program of FindAllSubmatch, nil: not served
	ng  int            // capture groups incl. group 0
}

func init() {
	// A binding compiled against another header must not run (cxg_timing grew in round 4, cxg_path_state_t in round 5).
	if C.cxg_abi_version() != C.CXG_ABI_VERSION {
		panic("coregex_hip: library ABI differs from the header this file was compiled against")
	}
}

// The library keeps its per-call scratch — a HIP stream, pinned control words, up to 256 MiB of HBM staging for host haystacks and
// rows — per OS THREAD (thread_local: the SearchState analogue), and a search must start and finish on one thread.  Goroutines
// migrate between the runtime's threads, and the runtime grows its thread count under blocking cgo calls: called from arbitrary
// goroutines the library would leave a scratch block behind on every M the runtime ever used.  So every search entry runs on one
// of a FIXED set of worker goroutines, each locked to its OS thread for life; a worker that has been idle for hipIdleRelease hands its
// scratch back (cxg_thread_release) and takes it again on its next call.  hipWorkers bounds the device memory the binding can
// pin: hipWorkers x 256 MiB, whatever GOMAXPROCS and however many goroutines call FindAll.  (Calls of different workers take
// turns on the device inside the library — the scans are HBM-bound — so more workers than a few buy overlap of the PCIe copies only.)
const (
	hipWorkers     = 4
	hipIdleRelease = 30 * time.Second
)

var (
	hipPoolOnce sync.Once
	hipJobs     chan func()
)

func hipWorker() {
	runtime.LockOSThread() // never unlocked: the thread dies with the goroutine, and its thread_local scratch with it
	idle := time.NewTimer(hipIdleRelease)
	released := true
	for {
		select {
		case job := <-hipJobs:
			job()
			released = false
			if !idle.Stop() {
				select {
				case <-idle.C:
				default:
				}
			}
			idle.Reset(hipIdleRelease)
		case <-idle.C:
			if !released {
				C.cxg_thread_release()
				released = true
			}
			idle.Reset(hipIdleRelease)
		}
	}
}

// hipDo runs f on a worker thread and waits for it.  The haystack and result slices f hands to C stay reachable through the
// caller's frame for the duration (cgo pins what a call passes; nothing is retained past return).
func hipDo(f func()) {
	hipPoolOnce.Do(func() {
		hipJobs = make(chan func())
		for i := 0; i < hipWorkers; i++ {
			go hipWorker()
		}
	})
	done := make(chan struct{})
	hipJobs <- func() { f(); close(done) }
	<-done
}

// buildHipProgram runs once per Engine, at the end of CompileRegexp (hooks.patch, meta/compile.go).
func (e *Engine) buildHipProgram() {
	if C.cxg_device_count() <= 0 {
		return
	}
	var prog *C.cxg_program
	switch e.strategy {
	case UseCharClassSearcher:
		if e.charClassSearcher == nil {
			return
		}
		member := e.charClassSearcher.Membership() // accessor added by nfa/charclass_searcher_hip.go
		var m [256]C.uint8_t
		for b := 0; b < 256; b++ {
			if member[b] {
				m[b] = 1
			}
		}
		minMatch := e.charClassSearcher.MinMatch()
		if minMatch < 1 { // `[class]*`: nullable — the library wants the NFA for those (cxg_program_from_nfa below)
			prog = e.nfaProgram(C.int(UseNFA), false)
			break
		}
		if C.cxg_program_from_charclass(&m[0], C.uint32_t(minMatch), &prog) != C.CXG_OK {
			prog = nil
		}
	case UseTeddy:
		pats := teddyPatterns(e.prefilter)
		if pats == nil { // (?m)^ wrapper or an unknown prefilter type: the NFA, strategy unchanged (the library checks the StartLine looks)
			prog = e.nfaProgram(C.int(UseTeddy), false)
			break
		}
		prog = literalsProgram(pats)
	case UseDigitPrefilter, UseDFA, UseBoth, UseNFA:
		prog = e.nfaProgram(C.int(e.strategy), false)
	case UseBoundedBacktracker:
		if !e.nfa.IsAlwaysAnchored() { // class-only patterns (`\S+`, `[0-9a-f]{32}`); anchored ones have no device kernel
			prog = e.nfaProgram(C.int(e.strategy), false)
		}
	default:
		return // reverse / composite / anchored strategies: CPU only
	}
	if prog != nil && C.cxg_program_supported(prog) != 1 {
		C.cxg_program_destroy(prog)
		prog = nil
	}
	// FindAllSubmatch: the same NFA with its real capture count (span program + capture table).  For NFAs with assertions
	// this is independent of `prog`: FindAllSubmatch of these strategies is the PikeVM (meta/findall.go:89-98).
	var sub *C.cxg_program
	if e.nfa.CaptureCount() > 1 {
		if s := e.nfaProgram(C.int(e.strategy), true); s != nil {
			if C.cxg_program_submatch_supported(s) == 1 {
				sub = s
			} else {
				C.cxg_program_destroy(s)
			}
		}
	}
	if prog == nil && sub == nil {
		return
	}
	h := &hipProgram{p: prog, sub: sub, ng: e.nfa.CaptureCount()}
	runtime.SetFinalizer(h, func(h *hipProgram) {
		if h.p != nil {
			C.cxg_program_destroy(h.p)
		}
		if h.sub != nil {
			C.cxg_program_destroy(h.sub)
		}
	})
	e.hip = h
}

// teddyPatterns: the literals of a Slim / Fat Teddy prefilter in pattern-ID order, nil for anything else
// (a lineAnchorWrapper around it, Aho-Corasick, memmem).
func teddyPatterns(p prefilter.Prefilter) [][]byte {
	switch t := p.(type) {
	case *prefilter.Teddy:
		return t.Patterns() // accessor added by prefilter/teddy_hip.go
	case *prefilter.FatTeddy:
		return t.Patterns()
	}
	return nil
}

// literalsProgram hands the literals over in C memory (cgo rule: memory passed to C must not hold Go pointers).
func literalsProgram(pats [][]byte) *C.cxg_program {
	n := len(pats)
	if n == 0 {
		return nil
	}
	ptrs := (**C.uint8_t)(C.malloc(C.size_t(n) * C.size_t(unsafe.Sizeof(uintptr(0)))))
	lens := (*C.uint32_t)(C.malloc(C.size_t(n) * 4))
	ps := unsafe.Slice(ptrs, n)
	ls := unsafe.Slice(lens, n)
	for i, p := range pats {
		ps[i] = (*C.uint8_t)(C.CBytes(p))
		ls[i] = C.uint32_t(len(p))
	}
	var prog *C.cxg_program
	rc := C.cxg_program_from_literals(ptrs, lens, C.uint32_t(n), &prog)
	for i := range pats {
		C.free(unsafe.Pointer(ps[i]))
	}
	C.free(unsafe.Pointer(ptrs))
	C.free(unsafe.Pointer(lens))
	if rc != C.CXG_OK {
		return nil
	}
	return prog
}

// flattenNFA copies e.nfa state by state through the exported accessors of nfa.State (nfa/nfa.go:157-233).  Kinds share
// their numbering with cxg_nfa_kind (nfa/nfa.go:23-60).  ok == false: the NFA holds a state kind the library does not
// take (StateRuneAny / StateRuneAnyNotNL: only the rune NFA of the PikeVM has them, never e.nfa — checked anyway).
func flattenNFA(n *nfa.NFA) (states []C.cxg_nfa_state, trans []C.cxg_nfa_trans, ok bool) {
	count := n.States()
	states = make([]C.cxg_nfa_state, count)
	invalid := C.uint32_t(C.CXG_NFA_INVALID)
	id32 := func(id nfa.StateID) C.uint32_t {
		if id == nfa.InvalidState {
			return invalid
		}
		return C.uint32_t(id)
	}
	for i := 0; i < count; i++ {
		s := n.State(nfa.StateID(i))
		c := &states[i]
		c.next, c.left, c.right = invalid, invalid, invalid
		switch s.Kind() {
		case nfa.StateMatch:
			c.kind = C.CXG_NFA_MATCH
		case nfa.StateByteRange:
			lo, hi, next := s.ByteRange()
			c.kind, c.lo, c.hi, c.next = C.CXG_NFA_BYTE_RANGE, C.uint8_t(lo), C.uint8_t(hi), id32(next)
		case nfa.StateSparse:
			ts := s.Transitions()
			c.kind = C.CXG_NFA_SPARSE
			c.trans_off, c.trans_len = C.uint32_t(len(trans)), C.uint32_t(len(ts))
			for _, t := range ts {
				trans = append(trans, C.cxg_nfa_trans{lo: C.uint8_t(t.Lo), hi: C.uint8_t(t.Hi), next: id32(t.Next)})
			}
		case nfa.StateSplit:
			l, r := s.Split()
			c.kind, c.left, c.right = C.CXG_NFA_SPLIT, id32(l), id32(r)
		case nfa.StateEpsilon:
			c.kind, c.next = C.CXG_NFA_EPSILON, id32(s.Epsilon())
		case nfa.StateCapture:
			idx, isStart, next := s.Capture()
			c.kind, c.cap_index, c.next = C.CXG_NFA_CAPTURE, C.uint32_t(idx), id32(next)
			if isStart {
				c.cap_start = 1
			}
		case nfa.StateFail:
			c.kind = C.CXG_NFA_FAIL
		case nfa.StateLook:
			look, next := s.Look()
			c.kind, c.lo, c.next = C.CXG_NFA_LOOK, C.uint8_t(look), id32(next) // nfa.Look numbering (nfa/nfa.go:92-117)
		default: // StateRuneAny (8), StateRuneAnyNotNL (9): the library answers CXG_E_UNSUPPORTED for them; do not even ask
			return nil, nil, false
		}
	}
	return states, trans, true
}

// nfaProgram hands e.nfa to cxg_program_from_nfa.  The cxg_nfa struct and both arrays are C memory: a struct holding the
// addresses of Go slices must not be passed to C, and the library keeps nothing past return.
func (e *Engine) nfaProgram(strategy C.int, captures bool) *C.cxg_program {
	states, trans, ok := flattenNFA(e.nfa)
	if !ok || len(states) == 0 {
		return nil
	}
	n := (*C.cxg_nfa)(C.calloc(1, C.size_t(unsafe.Sizeof(C.cxg_nfa{}))))
	defer C.free(unsafe.Pointer(n))
	sbytes := C.size_t(len(states)) * C.size_t(unsafe.Sizeof(states[0]))
	n.states = (*C.cxg_nfa_state)(C.malloc(sbytes))
	C.memcpy(unsafe.Pointer(n.states), unsafe.Pointer(&states[0]), sbytes)
	defer C.free(unsafe.Pointer(n.states))
	n.n_states = C.uint32_t(len(states))
	if len(trans) > 0 {
		tbytes := C.size_t(len(trans)) * C.size_t(unsafe.Sizeof(trans[0]))
		n.trans = (*C.cxg_nfa_trans)(C.malloc(tbytes))
		C.memcpy(unsafe.Pointer(n.trans), unsafe.Pointer(&trans[0]), tbytes)
		defer C.free(unsafe.Pointer(n.trans))
		n.n_trans = C.uint32_t(len(trans))
	}
	n.start_anchored = C.uint32_t(e.nfa.StartAnchored())
	n.start_unanchored = C.uint32_t(e.nfa.StartUnanchored())
	n.capture_count = 1 // FindAllIndex / Count: group 0 only
	if captures {
		n.capture_count = C.uint32_t(e.nfa.CaptureCount())
	}
	var flags C.uint32_t
	if e.digitRunSkipSafe { // meta/compile.go:176
		flags |= C.CXG_FLAG_DIGIT_RUN_SKIP_SAFE
	}
	if e.reverseDFA != nil { // meta/compile.go:184-205
		flags |= C.CXG_FLAG_HAS_REVERSE_DFA
	}
	if e.prefilter != nil { // UseBoth restarts its PikeVM at end-100 only without a prefilter (meta/find_indices.go:408-441)
		flags |= C.CXG_FLAG_HAS_PREFILTER
	}
	var prog *C.cxg_program
	if C.cxg_program_from_nfa(n, strategy, flags, &prog) != C.CXG_OK { // malformed input is CXG_E_INVALID, never a crash
		return nil
	}
	return prog
}

// bumpStats: once per batch, what the CPU loop would have counted per search is not known here — the batch counts as ONE search
// of the engine family that would have run (meta/engine.go:159-183; SURVEY section 5): DFASearches += 1 for the DFA strategies,
// NFASearches += 1 for UseNFA / UseBoundedBacktracker and for every FindAllSubmatch batch (the CPU path of those is the PikeVM,
// meta/findall.go:89-98), and for the prefilter strategies PrefilterHits += rows — the CPU loop counts a hit per candidate that
// verified, which for Teddy and the digit prefilter is once per match.  PrefilterMisses, the cache-fill counters and the
// per-search fallback counters stay untouched: nothing on the device corresponds to them.  A host that reads Stats to decide
// about engines sees the batches as cheap DFA searches, which is what they are from its side.
func (e *Engine) bumpHipStats(rows int) {
	switch e.strategy {
	case UseNFA, UseBoundedBacktracker:
		atomic.AddUint64(&e.stats.NFASearches, 1)
	case UseTeddy, UseDigitPrefilter:
		atomic.AddUint64(&e.stats.PrefilterHits, uint64(rows))
		atomic.AddUint64(&e.stats.DFASearches, 1)
	default:
		atomic.AddUint64(&e.stats.DFASearches, 1)
	}
}

// findAllHip is the batch path of FindAllIndicesStreaming.  ok == false: the caller runs the CPU loop.
func (e *Engine) findAllHip(haystack []byte, n int, results [][2]int) ([][2]int, bool) {
	if e.hip == nil || e.hip.p == nil || len(haystack) < hipThreshold {
		return nil, false
	}
	limit := C.int64_t(-1)
	if n > 0 {
		limit = C.int64_t(n)
	}
	if cap(results) == 0 {
		results = make([][2]int, 0, len(haystack)/100+1) // the reference's own first guess (meta/findall.go:190-200)
	}
	for {
		var got C.uint64_t
		results = results[:cap(results)]
		var rc C.int
		hipDo(func() {
			rc = C.cxg_find_all(e.hip.p, (*C.uint8_t)(unsafe.Pointer(&haystack[0])), C.uint64_t(len(haystack)), limit,
				(*C.int64_t)(unsafe.Pointer(&results[0])), C.uint64_t(len(results)), &got) // [2]int is int64[2] on the 64-bit targets the library exists for
		})
		switch rc {
		case C.CXG_OK:
			e.bumpHipStats(int(got))
			return results[:got], true
		case C.CXG_E_CAPACITY: // got = rows required: grow and retry
			results = make([][2]int, 0, int(got))
		default: // CXG_E_UNSUPPORTED, CXG_E_INPUT (this haystack only), CXG_E_NO_GPU, a device error: degrade, never fail
			return nil, false
		}
	}
}

// findHip is the device path of Find (meta/find.go:29): the first match of a whole haystack — FindAll with n == 1 on the device,
// whose early stop lets the workgroups behind the first counted row leave at once.  Worth it for haystacks whose first match is far
// in (or absent); a match in the first few KiB is found faster by the CPU search, which is why the threshold is the batch path's.
func (e *Engine) findHip(haystack []byte) (*Match, bool) {
	if e.hip == nil || e.hip.p == nil || len(haystack) < hipThreshold {
		return nil, false
	}
	var span [2]C.int64_t
	var found, rc C.int
	hipDo(func() {
		rc = C.cxg_find(e.hip.p, (*C.uint8_t)(unsafe.Pointer(&haystack[0])), C.uint64_t(len(haystack)), &span[0], &found)
	})
	if rc != C.CXG_OK {
		return nil, false
	}
	e.bumpHipStats(int(found))
	if found == 0 {
		return nil, true
	}
	return NewMatch(int(span[0]), int(span[1]), haystack), true
}

// isMatchHip is the device path of IsMatch (meta/ismatch.go:27).
func (e *Engine) isMatchHip(haystack []byte) (matched bool, ok bool) {
	if e.hip == nil || e.hip.p == nil || len(haystack) < hipThreshold {
		return false, false
	}
	var m, rc C.int
	hipDo(func() {
		rc = C.cxg_is_match(e.hip.p, (*C.uint8_t)(unsafe.Pointer(&haystack[0])), C.uint64_t(len(haystack)), &m)
	})
	if rc != C.CXG_OK {
		return false, false
	}
	e.bumpHipStats(int(m))
	return m != 0, true
}

// countHip is the batch path of Count (meta/findall.go:297).
func (e *Engine) countHip(haystack []byte, n int) (int, bool) {
	if e.hip == nil || e.hip.p == nil || len(haystack) < hipThreshold {
		return 0, false
	}
	limit := C.int64_t(-1)
	if n > 0 {
		limit = C.int64_t(n)
	}
	var got C.uint64_t
	var rc C.int
	hipDo(func() {
		rc = C.cxg_count(e.hip.p, (*C.uint8_t)(unsafe.Pointer(&haystack[0])), C.uint64_t(len(haystack)), limit, &got)
	})
	if rc != C.CXG_OK {
		return 0, false
	}
	e.bumpHipStats(int(got))
	return int(got), true
}

// findAllSubmatchHip is the batch path of FindAllSubmatch (meta/findall.go:390): rows of 2*groups int64, -1 for a group
// that did not take part, wrapped exactly as slotsToCaptures does (meta/findall.go:132-147).
func (e *Engine) findAllSubmatchHip(haystack []byte, n int) ([]*MatchWithCaptures, bool) {
	if e.hip == nil || e.hip.sub == nil || len(haystack) < hipThreshold {
		return nil, false
	}
	limit := C.int64_t(-1)
	if n > 0 {
		limit = C.int64_t(n)
	}
	width := 2 * e.hip.ng
	rows := make([]int64, (len(haystack)/100+1)*width)
	for {
		var got C.uint64_t
		var rc C.int
		hipDo(func() {
			rc = C.cxg_find_all_submatch(e.hip.sub, (*C.uint8_t)(unsafe.Pointer(&haystack[0])), C.uint64_t(len(haystack)), limit,
				(*C.int64_t)(unsafe.Pointer(&rows[0])), C.uint64_t(len(rows)/width), &got)
		})
		switch rc {
		case C.CXG_OK:
			out := make([]*MatchWithCaptures, int(got))
			for i := range out {
				slots := make([]int, width)
				for k := 0; k < width; k++ {
					slots[k] = int(rows[i*width+k])
				}
				out[i] = NewMatchWithCaptures(haystack, slotsToCaptures(slots))
			}
			atomic.AddUint64(&e.stats.NFASearches, 1)
			return out, true
		case C.CXG_E_CAPACITY:
			rows = make([]int64, int(got)*width)
		default:
			return nil, false
		}
	}
}
