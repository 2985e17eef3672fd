// TEST INFRASTRUCTURE ONLY — sanitizer driver for the HOST side of the library (front-end, program builder, transducer
// builder) and the sequential twin of the transducer kernel.  Built by `make san` with g++ -fsanitize=address,undefined;
// reads one pattern per line on stdin, runs the steps cxg_compile runs (capi.hip) and the twin over a few haystacks.
// Nothing in coregex_amd/ links it; no oracle involved (this looks for memory / UB errors, not for wrong answers).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <string>
#include <vector>

#include "../../coregex_amd/csrc/host/program.h"

extern "C" int64_t emu_captures_bt(const uint8_t* cap_blob, const uint8_t* hay, uint64_t len, const int64_t* spans, int64_t nrows, int64_t* out);
extern "C" int64_t emu_find_all_fsm(const uint8_t* img, const uint8_t* hay, uint64_t len, int64_t* out, int64_t cap_vals, int tile, int chunk,
                                    int budget_bytes, uint64_t* stats, int dense);

extern "C" const char* cxg_strategy_name(int) { return "strategy"; }   // capi.hip's name table is not linked here

int main() {
  std::string line;
  size_t n = 0, nprog = 0, nimg = 0, nrun = 0, nrejected = 0, naccepted = 0, ncap = 0;
  uint64_t seed = 0x9E3779B97F4A7C15ull;
  auto rnd = [&]() { seed ^= seed << 13; seed ^= seed >> 7; seed ^= seed << 17; return seed; };
  const char alphabet[] = "abcxyz.:-0123456789 \n_A@";
  while (std::getline(std::cin, line)) {
    n++;
    cxg_program p;
    try {
      cxg::Ast ast = cxg::parsePattern(line);
      try { p.nfa = cxg::buildNfa(ast); } catch (const cxg::FrontendError&) { continue; }
      cxg::Plan plan = cxg::selectStrategy(ast, p.nfa);
      cxg_nfa view = p.nfa.view();
      try {
        switch (plan.strategy) {
          case CXG_USE_CHARCLASS_SEARCHER: cxg::buildProgramFromCharClass(&p, plan.membership, 1); break;
          case CXG_USE_TEDDY: {
            if (plan.lineStart && !plan.lineStartAll) break;
            if (plan.lineStart) { cxg::buildProgramFromNfa(&p, view, CXG_USE_TEDDY, 0); break; }
            std::vector<std::vector<uint8_t>> lits;
            for (auto& l : plan.prefixes) lits.push_back(l.bytes);
            cxg::buildProgramFromLiterals(&p, lits);
            break;
          }
          default: cxg::buildProgramFromNfa(&p, view, plan.strategy, plan.flags); break;
        }
        if (p.nfa.captureCount > 1) cxg::buildSubmatchProgram(&p, view, plan.strategy);
        cxg::Ast sur;
        std::vector<std::pair<int, int>> bounds;
        if (p.supported && p.nfa.captureCount == 1 && cxg::boundedSurrogate(ast, sur, bounds)) {
          try { cxg::HostNfa sn = cxg::buildNfa(sur); cxg::attachBoundedChain(&p, sn.view(), bounds); } catch (const cxg::FrontendError&) {}
        }
      } catch (const cxg::BuildError&) { continue; }
      nprog++;
      // A caller's NFA with damaged fields (cxg_program_from_nfa): validateNfa must reject what the builders cannot walk.
      for (int rep = 0; rep < 4; rep++) {
        cxg::HostNfa bad = p.nfa;
        if (bad.states.empty()) break;
        for (int k = 0, nk = 1 + static_cast<int>(rnd() % 3); k < nk; k++) {
          cxg_nfa_state& st = bad.states[rnd() % bad.states.size()];
          const uint32_t v = (rnd() % 3 == 0) ? static_cast<uint32_t>(rnd()) : static_cast<uint32_t>(rnd() % (bad.states.size() + 3));
          switch (rnd() % 8) {
            case 0: st.next = v; break;
            case 1: st.left = v; break;
            case 2: st.right = v; break;
            case 3: st.trans_off = v; break;
            case 4: st.trans_len = v; break;
            case 5: st.kind = static_cast<uint8_t>(v); break;
            case 6: st.lo = static_cast<uint8_t>(v); st.hi = static_cast<uint8_t>(rnd()); break;
            default: if (rnd() & 1) bad.startAnchored = v; else bad.startUnanchored = v; break;
          }
        }
        cxg_nfa bv = bad.view();
        if (rnd() % 5 == 0) bv.capture_count = static_cast<uint32_t>(rnd() % 40);
        std::string why;
        if (!cxg::validateNfa(bv, why)) { nrejected++; continue; }
        cxg_program q;
        try {
          cxg::buildProgramFromNfa(&q, bv, plan.strategy == CXG_USE_CHARCLASS_SEARCHER ? CXG_USE_DFA : plan.strategy, plan.flags);
          if (bv.capture_count > 1) cxg::buildSubmatchProgram(&q, bv, plan.strategy);
        } catch (const cxg::BuildError&) {}
        naccepted++;
      }
      for (const std::vector<uint8_t>* img : {&p.fsmBlob, &p.subFsmBlob}) {
        if (img->empty()) continue;
        nimg++;
        for (int rep = 0; rep < 3; rep++) {
          const size_t len = rep == 0 ? 0 : (rnd() % 900);
          std::vector<uint8_t> hay(len + 1);
          const size_t na = rep == 2 ? 4 : sizeof alphabet - 1;
          for (size_t i = 0; i < len; i++) hay[i] = static_cast<uint8_t>(alphabet[rnd() % na]);
          std::vector<int64_t> out(2 * (len + 2));
          uint64_t stats[8] = {0};
          int64_t nvals = 0;
          for (auto g : {std::pair<int, int>{64, 8}, {256, 16}, {3840, 32}}) {
            nvals = emu_find_all_fsm(img->data(), hay.data(), len, out.data(), static_cast<int64_t>(out.size()), g.first, g.second, 0, stats, 0);
            nrun++;
          }
          // the backtracking capture pass over the spans of the last run (exactly `len` bytes: an assertion state must not read past them)
          if (img == &p.subFsmBlob && nvals > 0 && nvals <= static_cast<int64_t>(out.size()) && p.capBlob.size() >= 4 && std::memcmp(p.capBlob.data(), "TBXC", 4) == 0) {
            std::vector<uint8_t> exact(hay.begin(), hay.begin() + static_cast<long>(len));
            std::vector<int64_t> rows(static_cast<size_t>(nvals / 2) * 2 * p.nfa.captureCount);
            emu_captures_bt(p.capBlob.data(), exact.data(), len, out.data(), nvals / 2, rows.data());
            ncap++;
          }
        }
      }
    } catch (const cxg::FrontendError&) {
    }
  }
  std::printf("%zu patterns, %zu programs, %zu transducer images, %zu twin runs, %zu capture passes, damaged NFAs: %zu rejected by validateNfa, %zu built: no sanitizer report\n",
              n, nprog, nimg, nrun, ncap, nrejected, naccepted);
  return 0;
}
