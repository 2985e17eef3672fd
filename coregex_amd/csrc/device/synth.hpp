// synthlog-v1: the deterministic synthetic corpus of SURVEY §8(d).
//
// ASCII, '\n'-terminated log lines, generated per 4 KiB page from splitmix64(seed, page) so that any
// page is reproducible on the CPU and on the GPU without generating its predecessors.  Pages hold
// whole lines; the tail of a page is space-padded and closed with '\n'.  Line templates are modelled
// on the reference's differential corpus (meta/stdlib_compat_test.go:147-189): access-log line with an
// IPv4 address, timestamp, request, status and size; application-log line with a level word; key=value
// line; e-mail line; plus decoys for each benchmark pattern (`1.2.3`, `1..2.3.4`, `a@b`, near-miss
// level words).  `config` selects the mix (BASELINE.json configs 1-5).
// Plain C++: compiled into the fill kernel and into the host twin used by tests and the CPU baseline.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define CXG_SYNTH_HD __host__ __device__
#else
#define CXG_SYNTH_HD
#endif

namespace cxgsynth {

constexpr uint32_t kPage = 4096;

struct Rng {
  uint64_t s;
  CXG_SYNTH_HD uint64_t next() {  // splitmix64
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
  }
  CXG_SYNTH_HD uint32_t below(uint32_t n) { return static_cast<uint32_t>((next() >> 32) * static_cast<uint64_t>(n) >> 32); }
};

struct Out {
  uint8_t* p;
  uint32_t n;
  CXG_SYNTH_HD void ch(char c) { p[n++] = static_cast<uint8_t>(c); }
  CXG_SYNTH_HD void str(const char* s) { while (*s) p[n++] = static_cast<uint8_t>(*s++); }
  CXG_SYNTH_HD void num(uint32_t v) {
    char t[10]; int k = 0;
    do { t[k++] = static_cast<char>('0' + v % 10); v /= 10; } while (v);
    while (k) p[n++] = static_cast<uint8_t>(t[--k]);
  }
  CXG_SYNTH_HD void num2(uint32_t v) { ch(static_cast<char>('0' + (v / 10) % 10)); ch(static_cast<char>('0' + v % 10)); }
};

CXG_SYNTH_HD inline const char* word(uint32_t i) {
  switch (i & 31) {
    case 0: return "index"; case 1: return "users"; case 2: return "login"; case 3: return "static"; case 4: return "api";
    case 5: return "data"; case 6: return "config"; case 7: return "session"; case 8: return "profile"; case 9: return "images";
    case 10: return "report"; case 11: return "health"; case 12: return "search"; case 13: return "cart"; case 14: return "order";
    case 15: return "admin"; case 16: return "backup"; case 17: return "notes"; case 18: return "style"; case 19: return "main";
    case 20: return "vendor"; case 21: return "assets"; case 22: return "upload"; case 23: return "export"; case 24: return "status";
    case 25: return "metrics"; case 26: return "queue"; case 27: return "worker"; case 28: return "cache"; case 29: return "batch";
    case 30: return "token"; default: return "item";
  }
}
// The 16 prefix-free literals of BASELINE config 3 (Slim Teddy, 2 per bucket).
CXG_SYNTH_HD inline const char* level(uint32_t i) {
  switch (i & 15) {
    case 0: return "error"; case 1: return "warning"; case 2: return "fatal"; case 3: return "critical";
    case 4: return "panic"; case 5: return "timeout"; case 6: return "refused"; case 7: return "denied";
    case 8: return "googlebot"; case 9: return "bingbot"; case 10: return "yandexbot"; case 11: return "crawler";
    case 12: return "spider"; case 13: return "failure"; case 14: return "exception"; default: return "overflow";
  }
}
CXG_SYNTH_HD inline const char* nearmiss(uint32_t i) {  // share 2-byte fingerprints with level(), never equal
  switch (i & 7) {
    case 0: return "errand"; case 1: return "warden"; case 2: return "father"; case 3: return "crisp";
    case 4: return "pants"; case 5: return "timer"; case 6: return "refund"; default: return "dental";
  }
}
CXG_SYNTH_HD inline const char* month(uint32_t i) {
  switch (i % 12) {
    case 0: return "Jan"; case 1: return "Feb"; case 2: return "Mar"; case 3: return "Apr"; case 4: return "May"; case 5: return "Jun";
    case 6: return "Jul"; case 7: return "Aug"; case 8: return "Sep"; case 9: return "Oct"; case 10: return "Nov"; default: return "Dec";
  }
}
CXG_SYNTH_HD inline const char* tld(uint32_t i) {
  switch (i & 3) { case 0: return "com"; case 1: return "org"; case 2: return "net"; default: return "io"; }
}

CXG_SYNTH_HD inline void ip(Out& o, Rng& r) {
  o.num(r.below(256)); o.ch('.'); o.num(r.below(256)); o.ch('.'); o.num(r.below(256)); o.ch('.'); o.num(r.below(256));
}

// One line without the trailing '\n'; at most 200 bytes.
CXG_SYNTH_HD inline void line(Out& o, Rng& r, uint32_t config) {
  uint32_t t = r.below(100);
  // per-config mix: [access, app-log, key=value, e-mail, words]
  uint32_t a, b, c, d;
  switch (config) {
    case 3: a = 35; b = 80; c = 90; d = 95; break;    // level words ~1 per 200 B
    case 4: a = 25; b = 45; c = 60; d = 70; break;    // natural word / punctuation mix
    case 5: a = 30; b = 45; c = 55; d = 95; break;    // e-mail lines ~1 address per 150 B
    default: a = 70; b = 85; c = 95; d = 98; break;   // configs 1-2: access log, ~1 IPv4 per 100 B
  }
  if (t < a) {
    ip(o, r);
    o.str(" - - [");
    o.num2(1 + r.below(28)); o.ch('/'); o.str(month(r.below(12))); o.str("/2024:");
    o.num2(r.below(24)); o.ch(':'); o.num2(r.below(60)); o.ch(':'); o.num2(r.below(60));
    o.str(" +0000] \"");
    o.str(r.below(4) ? "GET /" : "POST /");
    o.str(word(r.below(32)));
    if (r.below(2)) { o.ch('/'); o.str(word(r.below(32))); }
    o.str(r.below(3) ? ".html" : ".php");
    o.str(" HTTP/1.1\" ");
    o.num(r.below(5) ? 200 : (r.below(2) ? 404 : 302));
    o.ch(' ');
    o.num(r.below(100000));
  } else if (t < b) {
    o.num2(r.below(24)); o.ch(':'); o.num2(r.below(60)); o.ch(':'); o.num2(r.below(60)); o.ch(' ');
    uint32_t k = r.below(10);
    if (k < 6) { o.ch('['); o.str(level(r.below(16))); o.str("] "); }
    else if (k < 8) { o.ch('['); o.str(nearmiss(r.below(8))); o.str("] "); }
    else o.str("[info] ");
    o.str(word(r.below(32))); o.ch(' '); o.str(word(r.below(32)));
    uint32_t dk = r.below(6);
    if (dk == 0) { o.str(" v"); o.num(r.below(10)); o.ch('.'); o.num(r.below(20)); o.ch('.'); o.num(r.below(100)); }          // 1.2.3
    else if (dk == 1) { o.ch(' '); o.num(r.below(10)); o.str(".."); o.num(r.below(256)); o.ch('.'); o.num(r.below(256)); o.ch('.'); o.num(r.below(256)); }  // 1..2.3.4
    else if (dk == 2) { o.str(" from "); ip(o, r); }
    o.str(" at line "); o.num(r.below(5000));
  } else if (t < c) {
    o.str("session_id="); for (int k = 0; k < 12; k++) o.ch("0123456789abcdef"[r.below(16)]);
    o.str(" user="); o.str(word(r.below(32))); o.num(r.below(1000));
    o.str(" ok="); o.num(r.below(2)); o.str(" ms="); o.num(r.below(3000));
  } else if (t < d) {
    uint32_t k = r.below(8);
    o.str(word(r.below(32))); o.ch(' ');
    if (k == 0) { o.str(word(r.below(32))); o.ch('@'); o.str(word(r.below(32))); }                       // a@b   (no dot)
    else if (k == 1) { o.ch('@'); o.str(word(r.below(32))); o.ch('.'); o.str(tld(r.below(4))); }        // @x.y
    else if (k == 2) { o.str(word(r.below(32))); o.ch('@'); o.str(word(r.below(32))); o.ch('.'); }      // a@b.
    else { o.str(word(r.below(32))); o.num(r.below(100)); o.ch('@'); o.str(word(r.below(32))); o.ch('.'); o.str(tld(r.below(4))); }
    o.str(" sent mail to "); o.str(word(r.below(32))); o.ch('_'); o.str(word(r.below(32))); o.ch('@'); o.str(word(r.below(32))); o.ch('.'); o.str(tld(r.below(4)));
  } else {
    uint32_t nw = 6 + r.below(10);
    for (uint32_t k = 0; k < nw; k++) {
      if (k) o.ch(r.below(8) ? ' ' : ',');
      if (r.below(6) == 0) o.num(r.below(100000)); else o.str(word(r.below(32)));
      if (r.below(12) == 0) { o.ch('_'); o.num(r.below(100)); }
    }
    o.ch('.');
  }
}

// Writes exactly kPage bytes.
CXG_SYNTH_HD inline void page(uint32_t config, uint64_t seed, uint64_t page_index, uint8_t* out) {
  Rng r{seed ^ (page_index * 0xD6E8FEB86659FD93ull + 0x2545F4914F6CDD1Dull)};
  r.next();
  uint32_t n = 0;
  uint8_t tmp[256];
  for (;;) {
    Out o{tmp, 0};
    line(o, r, config);
    if (n + o.n + 1 > kPage - 1) break;      // keep room for the closing '\n'
    for (uint32_t k = 0; k < o.n; k++) out[n + k] = tmp[k];
    n += o.n;
    out[n++] = '\n';
  }
  while (n < kPage - 1) out[n++] = ' ';
  out[n] = '\n';
}

}  // namespace cxgsynth
