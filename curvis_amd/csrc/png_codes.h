/* png_codes.h -- code lengths of an OPTIMAL length-limited prefix code (package-merge, Larmore & Hirschberg 1990), as plain
 * functions over caller-owned fixed arrays, for both sides of the PNG writers: the device builds the codes of every frame of a
 * deflate call in a kernel of its own (kernels_png.h png_codes_kernel: no host round trip between the histogram pass and the emit
 * pass), the host writer (host/png_io.h encode_rgb8_fast) calls pm_lengths_sorted below -- one definition of "the code of a
 * histogram", testable on the CPU.
 *
 * Package-merge in the form both sides use.  Leaves in non-decreasing weight order (ties by symbol index).  List 1 = the leaves;
 * list j = the leaves merged with the PACKAGES of list j - 1 (sums of its consecutive pairs, an odd last item dropped), a leaf
 * before a package of equal weight; the code takes the 2n - 2 cheapest items of list L (L = the length limit), every package
 * taken at level j takes its two items of list j - 1, and the length of a symbol is the number of levels at which its leaf is
 * taken.  Because every list is sorted, "taken" is a prefix at every level, so per level only the NUMBER of leaves in the prefix
 * is needed (pm_backtrack) -- no package carries its contents.  Merging two sorted lists is a rank computation (position = own
 * index + number of items of the other list that go before), which is what the device does with one lane per item; the host walks
 * the same lists with two pointers.  Weights are 64-bit: a package can outgrow 32 bits long before it could be taken.
 * (Rounds 3-6 halved the weights and rebuilt a Huffman tree until it was shallow enough: ~11 rebuilds for a 1080p frame, 60-80 us
 * per frame on a host core, 0.2 % above this optimum; the usual Kraft-sum repair of zlib-class encoders, tried first, was 3.5 %
 * above it on such histograms.) */
#pragma once
#include <stdint.h>

#if defined(__HIPCC__) || defined(__HIP__)
#define PNGC_HD __host__ __device__ inline
#else
#define PNGC_HD inline
#endif

namespace pngcodes {

constexpr int kMaxLeaves = 288; /* what the callers' arrays are sized for: 2 * kMaxLeaves - 1 nodes */
constexpr int kMaxLimit = 15;

/* what list j remembers of its items, one u16 per position: a leaf's index k, or kPmPackage | the package's index i */
constexpr unsigned kPmPackage = 0x8000u;

/* number of leaves among the first t items of a list (t >= 1) whose item t - 1 is `last`: all leaves up to index k if it is leaf
 * k, else the t - 1 items before package i are i packages and the rest leaves */
PNGC_HD unsigned pm_leaves_in_prefix(unsigned t, unsigned last) {
  return (last & kPmPackage) ? (t - 1u) - (last & (kPmPackage - 1u)) : last + 1u;
}
/* a[j] = leaves taken at level j (j = 1 .. limit) when the 2n - 2 cheapest items of list `limit` are taken;
 * src: [limit + 1][stride] (row 0 unused), row j = list j */
PNGC_HD void pm_backtrack(const uint16_t *src, unsigned stride, int n, int limit, unsigned *a) {
  unsigned t = 2u * (unsigned)n - 2u;
  for (int j = limit; j >= 1; --j) {
    if (t == 0u) {
      a[j] = 0u;
      continue;
    }
    a[j] = j == 1 ? t : pm_leaves_in_prefix(t, src[(size_t)j * stride + (t - 1u)]);
    t = 2u * (t - a[j]); /* every package taken here takes two items of the list below */
  }
}
PNGC_HD uint8_t pm_length(unsigned k, const unsigned *a, int limit) { /* leaf k of the sorted order */
  unsigned l = 0;
  for (int j = 1; j <= limit; ++j) l += k < a[j] ? 1u : 0u;
  return (uint8_t)l;
}

/* the host's walk: code lengths (<= limit <= kMaxLimit, 2^limit >= n) for n >= 1 leaves of weights wl[0..n) in non-decreasing
 * order -> len_sorted[k].  Scratch: two lists of 2n entries and limit + 1 rows of 2n u16. */
inline void pm_lengths_sorted(const uint32_t *wl, int n, int limit, uint8_t *len_sorted) {
  if (n == 1) {
    len_sorted[0] = 1;
    return;
  }
  const unsigned stride = 2u * kMaxLeaves;
  static thread_local uint64_t A[2][2 * kMaxLeaves];
  static thread_local uint16_t src[(kMaxLimit + 1) * 2 * kMaxLeaves];
  unsigned size = (unsigned)n;
  for (int k = 0; k < n; ++k) A[0][k] = wl[k];
  int cur = 0;
  for (int j = 2; j <= limit; ++j) {
    const uint64_t *P = A[cur];
    uint64_t *Q = A[cur ^ 1];
    const unsigned q = size / 2u;
    unsigned k = 0, i = 0, pos = 0;
    while (k < (unsigned)n || i < q) {
      const uint64_t pk = i < q ? P[2 * i] + P[2 * i + 1] : 0;
      if (k < (unsigned)n && (i >= q || (uint64_t)wl[k] <= pk)) { /* a leaf goes before a package of equal weight */
        Q[pos] = wl[k];
        src[(size_t)j * stride + pos] = (uint16_t)k;
        ++k;
      } else {
        Q[pos] = pk;
        src[(size_t)j * stride + pos] = (uint16_t)(kPmPackage | i);
        ++i;
      }
      ++pos;
    }
    size = pos;
    cur ^= 1;
  }
  unsigned a[kMaxLimit + 1];
  pm_backtrack(src, stride, n, limit, a);
  for (int k = 0; k < n; ++k) len_sorted[k] = pm_length((unsigned)k, a, limit);
}

/* literal/length symbol of a match length 3..258 (RFC 1951 3.2.5), its number of extra bits and their value */
PNGC_HD void length_symbol(int length, int &sym, int &ebits, int &eval) {
  if (length == 258) {
    sym = 285, ebits = 0, eval = 0;
    return;
  }
  if (length <= 10) {
    sym = 254 + length, ebits = 0, eval = 0;
    return;
  }
  /* 11..18: 1 extra bit, 19..34: 2, 35..66: 3, 67..130: 4, 131..257: 5 -- group g starts at 3 + (4 << g) and has four symbols of
   * 2^g lengths each */
  int g = 1;
  while (length >= 3 + (4 << (g + 1))) ++g;
  const int rel = length - (3 + (4 << g));
  sym = 261 + 4 * g + (rel >> g);
  ebits = g;
  eval = rel & ((1 << g) - 1);
}

/* canonical code (RFC 1951 3.2.2) of symbol i given every symbol's length: first[l] = smallest code of length l; the code is
 * first[len[i]] + the number of earlier symbols of the same length, returned bit-reversed for an LSB-first bit stream */
PNGC_HD void canonical_first(const unsigned *count /* [0..limit], count[0] ignored */, int limit, unsigned *first) {
  unsigned code = 0, prev = 0;
  first[0] = 0;
  for (int b = 1; b <= limit; ++b) {
    code = (code + prev) << 1;
    first[b] = code;
    prev = count[b];
  }
}
PNGC_HD unsigned reverse_bits(unsigned c, int l) {
  unsigned r = 0;
  for (int k = 0; k < l; ++k) r |= ((c >> k) & 1u) << (l - 1 - k);
  return r;
}

}  // namespace pngcodes
