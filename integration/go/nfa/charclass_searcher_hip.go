package nfa

// Accessors for the device binding (meta/findall_hip.go).  CharClassSearcher keeps its table private
// (nfa/charclass_searcher.go:21-27); the binding needs the 256 membership flags and minMatch once per compiled Engine.

// Membership returns a copy of the 256-entry membership table: Membership()[b] is true when byte b is in the class.
func (s *CharClassSearcher) Membership() [256]bool { return s.membership }

// MinMatch returns the minimum match length (1 for `+`, 0 for `*`).
func (s *CharClassSearcher) MinMatch() int { return s.minMatch }
