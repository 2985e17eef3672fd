//go:build !hip || !cgo

package meta

// Default build: no device path.  The hooks of hooks.patch compile to nothing.

type hipProgram struct{}

func (e *Engine) buildHipProgram() {}

func (e *Engine) findAllHip(haystack []byte, n int, results [][2]int) ([][2]int, bool) {
	return nil, false
}

func (e *Engine) countHip(haystack []byte, n int) (int, bool) { return 0, false }

func (e *Engine) findAllSubmatchHip(haystack []byte, n int) ([]*MatchWithCaptures, bool) {
	return nil, false
}

func (e *Engine) findHip(haystack []byte) (*Match, bool) { return nil, false }

func (e *Engine) isMatchHip(haystack []byte) (bool, bool) { return false, false }
