package prefilter

// Accessors for the device binding (meta/findall_hip.go).  Teddy and FatTeddy keep their literals private
// (prefilter/teddy.go:110-113, prefilter/teddy_fat.go:57-59); the binding hands them to cxg_program_from_literals in
// pattern-ID order, which is the order of verifyBucket (prefilter/teddy.go:532-550).  The slices are shared, not copied:
// callers must not modify them (the binding copies the bytes into C memory at once).

// Patterns returns the literals of a Slim Teddy in pattern-ID order.
func (t *Teddy) Patterns() [][]byte { return t.patterns }

// Patterns returns the literals of a Fat Teddy (33..64 patterns, 16 buckets) in pattern-ID order.
func (t *FatTeddy) Patterns() [][]byte { return t.patterns }
