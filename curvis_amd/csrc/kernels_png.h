/* kernels_png.h -- PNG front end on the device: the RGB8 frames a render call left in HBM -> one zlib stream per frame.
 * What the reference does with `image::DynamicImage::save` after every frame (src/rendering.rs:110, :311), done where the
 * frame already is, so that 6.2 MB of pixels per 1080p frame neither cross PCIe nor cost a host thread 4-7 ms of
 * filtering and Huffman coding (in `--mode efficient` the HOST was the limit of `curvis video`).
 *
 * Stream format (any PNG decoder reads it): filter type 2 (Up) on every row, ONE dynamic-Huffman deflate block, literals
 * and distance-1 matches for runs of zero bytes -- the format of the host writer (host/png_io.h encode_rgb8_fast) with one
 * difference: a thread tokenises 64 image bytes, so a zero run is cut every 64 bytes (matches of 3..63 instead of 3..258).
 *
 *   png_hist2_kernel      pass 1: token histogram (286 symbols) per frame AND per workgroup, Adler-32 partial sums
 *   png_codes_kernel      code lengths (<= 12 bits), canonical codes and the block header from the frame's histogram: a workgroup
 *                         per frame (png_codes.h: the host writer's functions).  Rounds 3-6 did this on the host, between two
 *                         synchronisations: 1.1-1.5 ms of a 128-frame call with the GPU idle
 *   png_blockbits_kernel  bits per workgroup = its token counts . bits per symbol (no pass over the pixels)
 *   png_offsets_kernel    exclusive scan of the workgroup totals of every frame, total stream length; headers; the few words
 *                         the emit pass ORs into are cleared
 *   png_emit2_kernel      pass 2: every thread counts its bits, the workgroup scans them, every thread writes its codes at its
 *                         bit offset -- assembled in LDS per workgroup (ds_or), written out as whole words (the first and last
 *                         word of a workgroup's span are shared with its neighbours: atomicOr)
 *   png_crc_kernel        CRC-32 of the PNG chunk ("IDAT" + stream)
 *
 * Byte work, no floating point.  TWO reads of the frames: the bits a workgroup's tokens take are  sum over the symbols of
 * count x bits(symbol), and pass 1 already counts the tokens -- so it also leaves its 288 counts PER WORKGROUP (576 B per
 * 16 KiB of pixels) and the bit offsets follow without a counting pass.  (Rounds 3-5 also carried the original three-pass
 * kernels -- histogram, count, emit, with a byte-serial tokeniser -- for frames whose rows are no multiple of 64 bytes and as the
 * checker of this path; round 6 taught the staging ragged rows (png_stage_ragged) and removed them: one path for every width.)
 * Each pass reads the frame once -- current row and row above, the latter out of the XCD's own L2 thanks to the workgroup order
 * (png_logical_block) --, stages the FILTERED bytes in LDS (png_stage: coalesced 16-byte loads when the rows are a multiple of
 * 64 bytes, i.e. every common video size) and tokenises 64 bytes per thread word-parallel: a thread holds its 64 filtered
 * bytes in 16 registers, classifies them four at a time (zero / +1 / -1 / other with carry-free byte tricks), builds the 64-bit
 * map of its zero bytes and walks the RUNS of that map (one iteration per run, not per byte).  The hottest symbols (literal 0,
 * 1, 255 and the match of a whole zero chunk) are counted in registers and reduced across the wave before they touch LDS.
 * Algorithmic bytes per frame: W*H*3 read + stream written.
 * Part of the ONE translation unit curvis_hip.hip (included there, nowhere else). */
#pragma once

namespace {

constexpr unsigned kPngChunk = 64;   /* image bytes per thread */
constexpr unsigned kPngBlock = 256;  /* threads per workgroup */
constexpr unsigned kPngBins = 288;   /* 286 literal/length symbols, padded */
constexpr unsigned kPngCodes = 256 + 64; /* per frame: literal entries [0, 256), match entries for lengths [0, 64) */
constexpr unsigned kPngCodeBits = 12;    /* longest literal/length code */
constexpr unsigned kPngHeaderWords = 44; /* zlib header (16 bits) + dynamic block header (<= 1222 bits) + slack: 176 bytes */

struct PngParams {
  const unsigned char *fb;        /* n_frames frames of H rows of row_bytes bytes, back to back */
  size_t frame_bytes;
  unsigned W, H, row_bytes, chunks_per_row, chunks_per_frame, blocks_per_frame, n_frames;
  int aligned;                    /* row_bytes % 4 == 0: a ragged frame's chunks are read four bytes at a time (else byte by byte) */
  int staged;                     /* row_bytes % 64 == 0: every chunk is full and a workgroup's 256 chunks are 16 KiB of consecutive
                                     bytes (png_stage); else the last chunk of a row is partial (png_stage_ragged) */
  unsigned grid_x;                /* 8 * ceil(blocks_per_frame / 8): see png_logical_block */
  unsigned *hist;                 /* [n_frames][kPngBins] */
  unsigned long long *adler;      /* [n_frames][2]: sum of the filtered bytes; sum of (n - i) * byte_i; both mod 65521 per workgroup */
  unsigned *codes;                /* [n_frames][kPngCodes]: bits | n_bits << 24 (png_codes_kernel) */
  unsigned long long *block_bits; /* [n_frames][blocks_per_frame]: bits per workgroup, then (in place) exclusive prefix */
  unsigned *start_bit;            /* [n_frames]: where the token stream starts (after the zlib and the block header) */
  unsigned long long *frame_bits; /* [n_frames]: end of the stream in bits (start offset, tokens, end-of-block code) */
  unsigned short *block_hist;     /* [n_frames][blocks_per_frame][kPngBins] token counts per workgroup (<= 16 641 each) */
  unsigned *sym_bits;             /* [n_frames][kPngBins] bits a token of each symbol takes (code + extra + distance) */
  unsigned *header;               /* [n_frames][kPngHeaderWords] zlib + block header bits (zero from start_bit on) */
  const unsigned *crc_tables;     /* [4][256] slice-by-4 tables of CRC-32 (reflected 0xEDB88320), then x^(8 d 16^i) mod p, [i = 0..7][d = 0..15] */
  unsigned *crc;                  /* [n_frames]: XOR of the threads' contributions = CRC state after "IDAT" + the frame's stream */
  unsigned *out;                  /* [n_frames][out_words] */
  size_t out_words;
};

/* length 3..66 -> literal/length symbol (RFC 1951 3.2.5) */
__device__ __forceinline__ unsigned png_len_symbol(unsigned len) {
  if (len <= 10u) return 254u + len;
  if (len <= 18u) return 265u + ((len - 11u) >> 1);
  if (len <= 34u) return 269u + ((len - 19u) >> 2);
  return 273u + ((len - 35u) >> 3);
}

/* bytes c - u, four at a time */
__device__ __forceinline__ unsigned png_sub4(unsigned c, unsigned u) {
  return ((c | 0x80808080u) - (u & 0x7f7f7f7fu)) ^ ((c ^ ~u) & 0x80808080u);
}

/* XCD-aware order of the workgroups.  The dispatcher deals consecutive workgroups round-robin to the 8 XCDs, each with an L2 of
 * its own; a workgroup reads the image rows of its chunks AND the row above them (filter Up), i.e. the rows of its
 * predecessor.  Workgroup b therefore takes the LOGICAL span (b mod 8) * ceil(n / 8) + b div 8: every XCD walks one contiguous
 * eighth of the frame and finds the row above in its own L2 (measured: FETCH_SIZE of a pass 3.1 x -> see profiles).  The grid
 * is 8 * ceil(n / 8) wide; spans >= n are idle. */
__device__ __forceinline__ unsigned png_logical_block(unsigned b, unsigned n_blocks) {
  const unsigned per = (n_blocks + 7u) >> 3;
  return (b & 7u) * per + (b >> 3);
}

/* Staging (frames whose rows are a multiple of 64 bytes, i.e. every chunk is full and the 256 chunks of a workgroup are 16 KiB
 * of consecutive image bytes): the workgroup loads its span and the span one row above with fully coalesced 16-byte loads
 * (lane i reads bytes [16 i, 16 i + 16) of a 4 KiB piece), subtracts, and leaves the FILTERED bytes in LDS, chunk c at
 * c * 80: the tokeniser reads its chunk 16 bytes at a time (ds_read_b128), and with 20 banks between neighbouring threads'
 * chunks the 16 lanes such a read serves per cycle hit 64 different banks. */
constexpr unsigned kPngLdsStride = 80;
__device__ __forceinline__ void png_stage(const PngParams &P, unsigned frame, unsigned block, unsigned char *s_f) {
  const unsigned char *f = P.fb + (size_t)frame * P.frame_bytes;
  const size_t base = (size_t)block * (kPngBlock * kPngChunk);
#pragma unroll
  for (unsigned k = 0; k < 4u; ++k) {
    /* 16-byte piece of the span.  A wave covers 64 consecutive pieces (1 KiB: the loads coalesce whatever the order inside);
     * lane L takes piece 4 (L mod 16) + L div 16 of them, so that the 16 lanes an LDS write serves together store the SAME
     * quarter of 16 DIFFERENT chunks -- bank offsets 20 c mod 64, a perfect cover -- instead of all quarters of 4 chunks, whose
     * fourth chunk lands on the banks of the first (round 5: the staging writes were the LDS bank conflicts that remained) */
    const unsigned j0 = threadIdx.x + kPngBlock * k, l = j0 & 63u;
    const unsigned j = (j0 & ~63u) + 4u * (l & 15u) + (l >> 4);
    const size_t off = base + (size_t)j * 16u;
    uint4 d = make_uint4(0u, 0u, 0u, 0u);
    if (off < P.frame_bytes) {
      const uint4 c = *reinterpret_cast<const uint4 *>(f + off);
      uint4 u = make_uint4(0u, 0u, 0u, 0u);
      if (off >= P.row_bytes) u = *reinterpret_cast<const uint4 *>(f + off - P.row_bytes);
      d = make_uint4(png_sub4(c.x, u.x), png_sub4(c.y, u.y), png_sub4(c.z, u.z), png_sub4(c.w, u.w));
    }
    *reinterpret_cast<uint4 *>(s_f + (j >> 2) * kPngLdsStride + (j & 3u) * 16u) = d;
  }
}

/* Ragged frames (rows that are no multiple of 64 bytes: the last chunk of every row is partial, chunks do not tile the frame):
 * every thread reads and filters its OWN chunk -- four bytes at a time when the rows are a multiple of 4 bytes, else byte by
 * byte -- and leaves it in its LDS slot, zero-padded to 64 bytes; the tokeniser masks the map of zero bytes to the chunk's
 * length (png_chunk_load), so the padding produces no token.  Uncoalesced (64-byte stride between lanes): such sizes are test
 * images and odd still frames, not video. */
__device__ __forceinline__ void png_stage_ragged(const PngParams &P, unsigned frame, unsigned block, unsigned char *s_f) {
  const unsigned g = block * kPngBlock + threadIdx.x;
  unsigned *dst = reinterpret_cast<unsigned *>(s_f + threadIdx.x * kPngLdsStride);
  unsigned nb = 0u;
  const unsigned char *cur = nullptr;
  bool has_up = false;
  if (g < P.chunks_per_frame) {
    const unsigned row = g / P.chunks_per_row, chunk = g - row * P.chunks_per_row, x0 = chunk * kPngChunk;
    nb = min(kPngChunk, P.row_bytes - x0);
    cur = P.fb + (size_t)frame * P.frame_bytes + (size_t)row * P.row_bytes + x0;
    has_up = row != 0u; /* row 0: the row above is treated as zeros, never read */
  }
  const unsigned char *up = cur - P.row_bytes; /* a POINTER one row up: an unsigned index `4 w + b - row_bytes` would wrap */
#pragma unroll 1
  for (unsigned w = 0; w < kPngChunk / 4u; ++w) {
    unsigned d = 0u;
    if (4u * w < nb) {
      if (P.aligned) { /* nb is a multiple of 4 then */
        const unsigned c = *reinterpret_cast<const unsigned *>(cur + 4u * w);
        d = png_sub4(c, has_up ? *reinterpret_cast<const unsigned *>(up + 4u * w) : 0u);
      } else {
        for (unsigned b = 0; b < 4u && 4u * w + b < nb; ++b)
          d |= (((unsigned)cur[4u * w + b] - (has_up ? (unsigned)up[4u * w + b] : 0u)) & 0xffu) << (8u * b);
      }
    }
    dst[w] = d;
  }
}
__device__ __forceinline__ void png_stage_any(const PngParams &P, unsigned frame, unsigned block, unsigned char *s_f) {
  if (P.staged) /* uniform over the launch */
    png_stage(P, frame, block, s_f);
  else
    png_stage_ragged(P, frame, block, s_f);
}
/* bytes of chunk `chunk` of a row (64, or what is left of a ragged row) */
__device__ __forceinline__ unsigned png_chunk_bytes(const PngParams &P, unsigned chunk) { return min(kPngChunk, P.row_bytes - chunk * kPngChunk); }

/* codes ORed into the workgroup's LDS image of the stream */
struct PngEmitSink {
  const unsigned *codes; /* LDS */
  unsigned *out;         /* LDS */
  unsigned w, fill;
  unsigned long long acc;
  __device__ __forceinline__ void put(unsigned e) {
    acc |= (unsigned long long)(e & 0xffffffu) << fill;
    fill += e >> 24;
    if (fill >= 32u) {
      atomicOr(&out[w], (unsigned)acc);
      ++w;
      acc >>= 32;
      fill -= 32u;
    }
  }
};

constexpr unsigned kPngHistReplicas = 4;   /* LDS histogram copies (lane & 3): same-value literals of neighbouring lanes do not collide */
constexpr unsigned kPngMatchReplicas = 16; /* copies of the 32 bins from 256 on (lane & 15): neighbouring chunks tend to hold runs of the SAME length */
constexpr unsigned kPngImageWords = 6272;  /* LDS image of a workgroup's piece of the stream in the emit pass: the worst case --
                                              65 literals of 12 bits per thread = 780 bits x 256 threads = 6240 words -- fits */
static_assert(kPngImageWords * 4u >= kPngBlock * kPngLdsStride, "the stream image re-uses the staging buffer");

/* bit 7 of every byte of x that is zero (exact: no carries between bytes) */
__device__ __forceinline__ unsigned png_zero_flags(unsigned x) { return ~(((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x | 0x7f7f7f7fu); }
/* the four flags (bits 7, 15, 23, 31) as a nibble: bits 0, 8, 16 gathered by one 24-bit multiply, bit 24 by a shift */
__device__ __forceinline__ unsigned png_flags_nibble(unsigned f) {
  return ((__umul24((f >> 7) & 0x00010101u, 0x00204081u) >> 21) & 7u) | ((f >> 28) & 8u);
}
__device__ __forceinline__ unsigned png_wave_sum(unsigned v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

/* a thread's chunk: 64 filtered bytes in registers + the map of its zero bytes */
struct PngChunk {
  unsigned d[16];
  unsigned long long Z; /* bit i: byte i is zero */
};
__device__ __forceinline__ void png_chunk_load(const unsigned char *s_chunk, unsigned nb, PngChunk &c) { /* nb: bytes of the chunk (1..64) */
#pragma unroll
  for (unsigned q = 0; q < 4u; ++q) {
    const uint4 v = *reinterpret_cast<const uint4 *>(s_chunk + 16u * q);
    c.d[4 * q + 0] = v.x, c.d[4 * q + 1] = v.y, c.d[4 * q + 2] = v.z, c.d[4 * q + 3] = v.w;
  }
  unsigned lo = 0u, hi = 0u;
#pragma unroll
  for (unsigned w = 0; w < 8u; ++w) lo |= png_flags_nibble(png_zero_flags(c.d[w])) << (4u * w);
#pragma unroll
  for (unsigned w = 0; w < 8u; ++w) hi |= png_flags_nibble(png_zero_flags(c.d[8u + w])) << (4u * w);
  c.Z = ((unsigned long long)hi << 32) | lo;
  if (nb < kPngChunk) c.Z &= (1ull << nb) - 1ull; /* ragged rows: the zero padding behind a partial chunk is not part of the stream */
}
/* length of the run of ones of Z that starts at bit s (Z has bit s set) */
__device__ __forceinline__ unsigned png_run_length(unsigned long long Z, unsigned s) {
  const unsigned long long t = ~(Z >> s); /* the run ends at the first zero bit of Z >> s; the shift brought zeros in from the top */
  return (unsigned)__ffsll((long long)t) - 1u;   /* s > 0: bit 64 - s of t is set.  s == 0 and Z all ones: t == 0 -> ffs = 0 -> handled by the caller */
}
/* zero runs of a chunk in stream order: f(start, length); a run of L <= 3 zeros is L literals, a longer one a literal zero and a
 * distance-1 match of L - 1 (3..63) -- the host writer's rule (host/png_io.h encode_rgb8_fast), cut at the chunk's end */
template <typename F>
__device__ __forceinline__ void png_for_runs(unsigned long long Z, F f) {
  if (Z == ~0ull) {
    f(0u, 64u);
    return;
  }
  while (Z) {
    const unsigned s = (unsigned)__ffsll((long long)Z) - 1u;
    const unsigned L = png_run_length(Z, s);
    f(s, L);
    Z = (s + L >= 64u) ? 0ull : (Z & (~0ull << (s + L)));
  }
}

/* pass 1: token histogram per frame AND per workgroup, Adler-32 partial sums */
__global__ __launch_bounds__(kPngBlock) void png_hist2_kernel(const PngParams P) {
  __shared__ unsigned s_hist[kPngHistReplicas][kPngBins];
  __shared__ unsigned s_match[kPngMatchReplicas][32];
  __shared__ unsigned long long s_sum[2];
  __shared__ __attribute__((aligned(16))) unsigned char s_f[kPngBlock * kPngLdsStride];
  const unsigned frame = blockIdx.y, block = png_logical_block(blockIdx.x, P.blocks_per_frame);
  if (block >= P.blocks_per_frame) return; /* the whole workgroup */
  const unsigned g = block * kPngBlock + threadIdx.x;
  for (unsigned k = threadIdx.x; k < kPngHistReplicas * kPngBins; k += kPngBlock) (&s_hist[0][0])[k] = 0u;
  for (unsigned k = threadIdx.x; k < kPngMatchReplicas * 32u; k += kPngBlock) (&s_match[0][0])[k] = 0u;
  if (threadIdx.x < 2u) s_sum[threadIdx.x] = 0ull;
  png_stage_any(P, frame, block, s_f);
  __syncthreads();
  unsigned *my_hist = s_hist[threadIdx.x & (kPngHistReplicas - 1u)];
  unsigned *my_match = s_match[threadIdx.x & (kPngMatchReplicas - 1u)];
  /* hot symbols in registers: literal 0, literal 1, literal 255, the match of a whole zero chunk (63) */
  unsigned n_zero = 0u, n_one = 0u, n_ff = 0u, n_m63 = 0u, a = 0u, b_mod = 0u;
  if (g < P.chunks_per_frame) {
    const unsigned row = g / P.chunks_per_row, chunk = g - row * P.chunks_per_row;
    PngChunk c;
    png_chunk_load(s_f + threadIdx.x * kPngLdsStride, png_chunk_bytes(P, chunk), c);
    const unsigned off = chunk == 0u ? 1u : 0u; /* chunk 0 of a row starts with the filter-type byte (2 = Up): stream index 0 */
    unsigned kv = 0u;
    if (off) {
      atomicAdd(&my_hist[2], 1u);
      a = 2u;
    }
    if (c.Z == ~0ull) { /* the common case after the Up filter: nothing but zeros */
      n_zero = 1u;
      n_m63 = 1u;
    } else {
#pragma unroll
      for (unsigned w = 0; w < 16u; ++w) {
        const unsigned x = c.d[w];
        if (x == 0u) continue;
        const unsigned f0 = png_zero_flags(x), f1 = png_zero_flags(x ^ 0x01010101u), ff = png_zero_flags(~x);
        n_one += __popc(f1);
        n_ff += __popc(ff);
        const unsigned sum = __builtin_amdgcn_sad_u8(x, 0u, 0u); /* the four bytes added up */
        a += sum;
        /* sum of (stream index of the byte) x byte: index = off + 4 w + position in the word */
        kv += (off + 4u * w) * sum + ((x >> 8) & 0xffu) + 2u * ((x >> 16) & 0xffu) + 3u * (x >> 24);
        unsigned other = 0x80808080u & ~(f0 | f1 | ff); /* bytes that are neither 0, 1 nor 255: through LDS, one by one */
        while (other) {
          const unsigned sh = (unsigned)__ffs((int)other) - 8u; /* flag at bit 8 j + 7 -> shift 8 j */
          atomicAdd(&my_hist[(x >> sh) & 0xffu], 1u);
          other &= other - 1u;
        }
      }
      png_for_runs(c.Z, [&](unsigned, unsigned L) {
        if (L <= 3u) {
          n_zero += L;
        } else {
          n_zero += 1u;
          atomicAdd(&my_match[png_len_symbol(L - 1u) - 256u], 1u);
        }
      });
    }
    /* Adler-32: s2 = n + sum over the stream of (n - i) * byte_i; this thread's bytes sit at i = i0 + k */
    const unsigned long long n = (unsigned long long)P.H * (P.row_bytes + 1u);
    const unsigned long long i0 = (unsigned long long)row * (P.row_bytes + 1u) + (chunk == 0u ? 0u : 1u + chunk * kPngChunk);
    b_mod = (unsigned)(((n - i0) * a - kv) % 65521ull);
  }
  /* one LDS atomic per wave and hot symbol instead of one per thread on the same address */
  const unsigned zero_one = png_wave_sum(n_zero | (n_one << 16)); /* <= 64 x 65 each: 16 bits are plenty */
  const unsigned ff_m63 = png_wave_sum(n_ff | (n_m63 << 16));
  const unsigned a_w = png_wave_sum(a), b_w = png_wave_sum(b_mod);
  if ((threadIdx.x & 63u) == 0u) {
    unsigned *h = s_hist[(threadIdx.x >> 6) & (kPngHistReplicas - 1u)];
    atomicAdd(&h[0], zero_one & 0xffffu);
    atomicAdd(&h[1], zero_one >> 16);
    atomicAdd(&h[255], ff_m63 & 0xffffu);
    atomicAdd(&h[png_len_symbol(63u)], ff_m63 >> 16);
    atomicAdd(&s_sum[0], (unsigned long long)a_w);
    atomicAdd(&s_sum[1], (unsigned long long)b_w);
  }
  __syncthreads();
  unsigned short *bh = P.block_hist + ((size_t)frame * P.blocks_per_frame + block) * kPngBins;
  for (unsigned k = threadIdx.x; k < kPngBins; k += kPngBlock) {
    unsigned v = 0u;
#pragma unroll
    for (unsigned r = 0; r < kPngHistReplicas; ++r) v += s_hist[r][k];
    if (k >= 256u)
      for (unsigned r = 0; r < kPngMatchReplicas; ++r) v += s_match[r][k - 256u];
    bh[k] = (unsigned short)v; /* <= 16 384 + 257 tokens per workgroup */
    if (v) atomicAdd(&P.hist[(size_t)frame * kPngBins + k], v);
  }
  if (threadIdx.x < 2u) atomicAdd(&P.adler[(size_t)frame * 2 + threadIdx.x], s_sum[threadIdx.x] % 65521ull);
}

/* The frame's code: lengths from the histogram (every symbol keeps a code: +1 on each count, +1 more for the end-of-block symbol --
 * which also makes the block header the host writer's fixed-size one), canonical codes, the table the emit pass indexes (literals,
 * end of block, matches of length 3..63 with extra bits and the one-bit distance code folded in), bits per token of every symbol for
 * png_blockbits_kernel, and the zlib + dynamic block header (16 + 1222 bits, RFC 1951 3.2.7: HLIT = 29, HDIST = 0, all 19
 * code-length codes of 3 bits declaring "4 bits" for lengths 0..15 and nothing for the repeat codes, then the 286 + 1 lengths raw).
 * One workgroup per frame, a lane per item: the symbols are ranked by (count, index), the eleven merges of package-merge
 * (png_codes.h) are rank computations by binary search, one lane walks the twelve levels back, the rest is per symbol again. */
constexpr unsigned kPngStartBit = 16u + 1222u;
constexpr unsigned kPngCodesThreads = 576; /* a lane per item of the longest list (286 leaves + 285 packages): one binary search per lane and level */
__global__ __launch_bounds__(kPngCodesThreads) void png_codes_kernel(const PngParams P) {
  constexpr unsigned kStride = 2u * pngcodes::kMaxLeaves;
  __shared__ unsigned s_freq[kPngBins], s_wl[kPngBins];
  __shared__ unsigned long long s_list[2][kStride];
  __shared__ unsigned short s_src[(kPngCodeBits + 1u) * kStride], s_order[kPngBins];
  __shared__ __attribute__((aligned(4))) unsigned char s_len[kPngBins];
  __shared__ unsigned s_taken[16], s_count[16], s_first[16], s_code[kPngBins], s_hdr[kPngHeaderWords];
  const unsigned frame = blockIdx.x;
  constexpr unsigned n = 286u;
  for (unsigned i = threadIdx.x; i < n; i += kPngCodesThreads) s_freq[i] = P.hist[(size_t)frame * kPngBins + i] + (i == 256u ? 2u : 1u);
  if (threadIdx.x < 16u) s_count[threadIdx.x] = 0u;
  for (unsigned i = threadIdx.x; i < kPngHeaderWords; i += kPngCodesThreads) s_hdr[i] = 0u;
  __syncthreads();
  for (unsigned i = threadIdx.x; i < n; i += kPngCodesThreads) s_list[1][i] = ((unsigned long long)s_freq[i] << 9) | i; /* (count, index) as one key */
  __syncthreads();
  for (unsigned i = threadIdx.x; i < n; i += kPngCodesThreads) { /* rank among the keys: they are distinct */
    const unsigned long long ki = s_list[1][i];
    unsigned rank = 0u;
#pragma unroll 2
    for (unsigned j = 0; j < n; ++j) rank += s_list[1][j] < ki ? 1u : 0u;
    s_order[rank] = (unsigned short)i;
    s_wl[rank] = s_freq[i];
    s_list[0][rank] = s_freq[i]; /* list 1: the leaves */
  }
  __syncthreads();
  /* lists 2 .. limit: the leaves merged with the packages of the list below, a leaf before a package of equal weight */
  unsigned size = n, cur = 0u;
  for (unsigned j = 2u; j <= kPngCodeBits; ++j) {
    const unsigned long long *A = s_list[cur];
    unsigned long long *B = s_list[cur ^ 1u];
    unsigned short *src = s_src + j * kStride;
    const unsigned q = size >> 1;
    for (unsigned t = threadIdx.x; t < n + q; t += kPngCodesThreads) {
      if (t < n) { /* leaf t: behind the packages that are strictly cheaper */
        const unsigned long long w = s_wl[t];
        unsigned lo = 0u, hi = q;
        while (lo < hi) {
          const unsigned mid = (lo + hi) >> 1;
          if (A[2u * mid] + A[2u * mid + 1u] < w)
            lo = mid + 1u;
          else
            hi = mid;
        }
        B[t + lo] = w;
        src[t + lo] = (unsigned short)t;
      } else { /* package i: behind the leaves that are no dearer */
        const unsigned i = t - n;
        const unsigned long long w = A[2u * i] + A[2u * i + 1u];
        unsigned lo = 0u, hi = n;
        while (lo < hi) {
          const unsigned mid = (lo + hi) >> 1;
          if ((unsigned long long)s_wl[mid] <= w)
            lo = mid + 1u;
          else
            hi = mid;
        }
        B[i + lo] = w;
        src[i + lo] = (unsigned short)(pngcodes::kPmPackage | i);
      }
    }
    size = n + q;
    cur ^= 1u;
    __syncthreads();
  }
  if (threadIdx.x == 0u) pngcodes::pm_backtrack(s_src, kStride, (int)n, (int)kPngCodeBits, s_taken);
  __syncthreads();
  for (unsigned k = threadIdx.x; k < n; k += kPngCodesThreads) {
    const unsigned l = pngcodes::pm_length(k, s_taken, (int)kPngCodeBits);
    s_len[s_order[k]] = (unsigned char)l;
    atomicAdd(&s_count[l], 1u);
  }
  __syncthreads();
  if (threadIdx.x == 0u) pngcodes::canonical_first(s_count, (int)kPngCodeBits, s_first);
  __syncthreads();
  for (unsigned i = threadIdx.x; i < n; i += kPngCodesThreads) { /* code = first of its length + earlier symbols of that length */
    const unsigned l = s_len[i];
    unsigned before = 0u;
    const unsigned *lw = reinterpret_cast<const unsigned *>(s_len); /* four lengths per word */
    const unsigned pat = l * 0x01010101u;
    for (unsigned j = 0; j < (i >> 2); ++j) before += (unsigned)__popc(png_zero_flags(lw[j] ^ pat));
    for (unsigned j = i & ~3u; j < i; ++j) before += s_len[j] == l ? 1u : 0u;
    s_code[i] = pngcodes::reverse_bits(s_first[l] + before, (int)l);
  }
  __syncthreads();
  unsigned *c = P.codes + (size_t)frame * kPngCodes;
  unsigned *sb = P.sym_bits + (size_t)frame * kPngBins;
  for (unsigned k = threadIdx.x; k < kPngCodes; k += kPngCodesThreads) {
    unsigned e = 0u;
    if (k <= 256u) { /* literals; 256: end of block, in the slot of the impossible "match of length 0" */
      e = s_code[k] | ((unsigned)s_len[k] << 24);
    } else if (k >= 256u + 3u) { /* a match of length k - 256 = 3..63: code, extra bits, the distance code (one zero bit) */
      int sym, eb, ev;
      pngcodes::length_symbol((int)(k - 256u), sym, eb, ev);
      const unsigned cl = s_len[sym];
      e = (s_code[sym] | ((unsigned)ev << cl)) | ((cl + (unsigned)eb + 1u) << 24);
    }
    c[k] = e;
  }
  for (unsigned i = threadIdx.x; i < kPngBins; i += kPngCodesThreads) {
    unsigned bits = 0u;
    if (i <= 256u) {
      bits = s_len[i];
    } else if (i < n) { /* length symbols 257..285: code + extra bits + distance code (those beyond length 63 never occur) */
      const unsigned g = i < 265u ? 0u : (i - 261u) / 4u;
      bits = (unsigned)s_len[i] + (i == 285u ? 0u : g) + 1u;
    }
    sb[i] = bits;
  }
  /* header: 0x78 0x01, then LSB first BFINAL = 1, BTYPE = 2, HLIT = 29, HDIST = 0, HCLEN = 15, the 19 three-bit code-length-code
   * lengths in the order 16 17 18 0 8 7 9 6 10 5 11 4 12 3 13 2 14 1 15 (0 for the three repeat codes, 4 for the rest), then
   * every symbol's length as a bit-reversed nibble (the canonical 4-bit code of the value), then the distance code's length 1 */
  auto put = [&](unsigned pos, unsigned v, unsigned nb) {
    atomicOr(&s_hdr[pos >> 5], v << (pos & 31u));
    if ((pos & 31u) + nb > 32u) atomicOr(&s_hdr[(pos >> 5) + 1u], v >> (32u - (pos & 31u)));
  };
  auto rev4 = [](unsigned v) { return ((v & 1u) << 3) | ((v & 2u) << 1) | ((v & 4u) >> 1) | ((v & 8u) >> 3); };
  if (threadIdx.x == 0u) {
    put(0u, 0x0178u, 16u);
    put(16u, 1u, 1u);
    put(17u, 2u, 2u);
    put(19u, 29u, 5u);
    put(24u, 0u, 5u);
    put(29u, 15u, 4u);
    for (unsigned k = 0; k < 19u; ++k) put(33u + 3u * k, k < 3u ? 0u : 4u, 3u);
    put(90u + 4u * n, rev4(1u), 4u);
  }
  for (unsigned i = threadIdx.x; i < n; i += kPngCodesThreads) put(90u + 4u * i, rev4(s_len[i]), 4u);
  __syncthreads();
  for (unsigned i = threadIdx.x; i < kPngHeaderWords; i += kPngCodesThreads) P.header[(size_t)frame * kPngHeaderWords + i] = s_hdr[i];
  if (threadIdx.x == 0u) P.start_bit[frame] = kPngStartBit;
}

/* bits per workgroup = its token counts . bits per symbol (+ the end-of-block code after the last one): one WAVE per
 * workgroup's 288 counts (36 lanes x 16 bytes), four per launch workgroup */
__global__ __launch_bounds__(kPngBlock) void png_blockbits_kernel(const PngParams P) {
  __shared__ unsigned s_bits[kPngBins];
  const unsigned frame = blockIdx.y, nb = P.blocks_per_frame;
  for (unsigned k = threadIdx.x; k < kPngBins; k += kPngBlock) s_bits[k] = P.sym_bits[(size_t)frame * kPngBins + k];
  __syncthreads();
  const unsigned lane = threadIdx.x & 63u, b = blockIdx.x * (kPngBlock / 64u) + (threadIdx.x >> 6);
  if (b >= nb) return; /* the whole wave */
  unsigned bits = 0u;
  if (lane < kPngBins / 8u) {
    const uint4 h = reinterpret_cast<const uint4 *>(P.block_hist + ((size_t)frame * nb + b) * kPngBins)[lane];
    const unsigned *sb = s_bits + 8u * lane;
    bits = (h.x & 0xffffu) * sb[0] + (h.x >> 16) * sb[1] + (h.y & 0xffffu) * sb[2] + (h.y >> 16) * sb[3] +
           (h.z & 0xffffu) * sb[4] + (h.z >> 16) * sb[5] + (h.w & 0xffffu) * sb[6] + (h.w >> 16) * sb[7];
  }
  bits = png_wave_sum(bits);
  if (lane == 0u) P.block_bits[(size_t)frame * nb + b] = bits + (b + 1u == nb ? s_bits[256] : 0u);
}

/* one workgroup per frame: block_bits -> exclusive prefix (in place), frame_bits = start + total; and the ONLY zeroing the
 * emit pass needs: a workgroup of png_emit2_kernel stores every word strictly inside its span and ORs into the first and
 * the last one, which it shares with its neighbours -- so the first word of every span and the last word of the stream are
 * cleared here, and the words before the first span receive the zlib and block headers
 * (a few KB per frame instead of a pass over the whole stream) */
__global__ __launch_bounds__(kPngBlock) void png_offsets_kernel(const PngParams P) {
  __shared__ unsigned long long s_part[kPngBlock];
  const unsigned frame = blockIdx.x, nb = P.blocks_per_frame;
  unsigned long long *v = P.block_bits + (size_t)frame * nb;
  unsigned *out = P.out + (size_t)frame * P.out_words;
  const unsigned start = P.start_bit[frame];
  const unsigned per = (nb + kPngBlock - 1u) / kPngBlock;
  const unsigned lo = min(nb, threadIdx.x * per), hi = min(nb, lo + per);
  unsigned long long sum = 0ull;
  for (unsigned k = lo; k < hi; ++k) sum += v[k];
  s_part[threadIdx.x] = sum;
  /* the zlib and block headers the host built from this frame's code lengths: bits [0, start); the word holding bit `start`
   * carries the header's last bits and is ORed into by the first span */
  for (unsigned k = threadIdx.x; k <= (start >> 5); k += kPngBlock) out[k] = k < kPngHeaderWords ? P.header[(size_t)frame * kPngHeaderWords + k] : 0u;
  __syncthreads();
  if (threadIdx.x == 0u) { /* 256 partial sums: a serial pass is as fast as anything here */
    unsigned long long run = 0ull;
    for (unsigned k = 0; k < kPngBlock; ++k) {
      const unsigned long long t = s_part[k];
      s_part[k] = run;
      run += t;
    }
    const unsigned long long end = (unsigned long long)start + run;
    P.frame_bits[frame] = end;
    if ((end >> 5) > (start >> 5)) out[end >> 5] = 0u; /* (never the word that holds the header's last bits) */
    out[(end >> 5) + 1u] = 0u;
  }
  __syncthreads();
  unsigned long long run = s_part[threadIdx.x];
  for (unsigned k = lo; k < hi; ++k) {
    const unsigned long long t = v[k];
    v[k] = run;
    /* first word of this workgroup's span (several tiny spans may share one: all clear it) -- except the word the header ends
     * in, which the first span ORs into: it was written above with the header's bits */
    if (((start + run) >> 5) > (start >> 5)) out[(start + run) >> 5] = 0u;
    run += t;
  }
}

/* pass 2: count this thread's bits, scan inside the workgroup, write the codes */
__global__ __launch_bounds__(kPngBlock) void png_emit2_kernel(const PngParams P) {
  __shared__ unsigned s_codes[kPngCodes];
  __shared__ unsigned s_wave[kPngBlock / 64];
  __shared__ __attribute__((aligned(16))) unsigned s_image[kPngImageWords]; /* first the staged filtered bytes, then the stream image */
  const unsigned frame = blockIdx.y, block = png_logical_block(blockIdx.x, P.blocks_per_frame);
  if (block >= P.blocks_per_frame) return;
  const unsigned g = block * kPngBlock + threadIdx.x;
  for (unsigned k = threadIdx.x; k < kPngCodes; k += kPngBlock) s_codes[k] = P.codes[(size_t)frame * kPngCodes + k];
  png_stage_any(P, frame, block, reinterpret_cast<unsigned char *>(s_image));
  __syncthreads();
  const bool live = g < P.chunks_per_frame;
  const bool first = live && (g % P.chunks_per_row) == 0u, last = g + 1u == P.chunks_per_frame;
  PngChunk c;
  c.Z = 0ull;
  unsigned long long ZL = 0ull, MS = 0ull; /* zero bytes that are emitted as literals; of those, the ones a match follows */
  unsigned mine = 0u;
  if (live) {
    png_chunk_load(reinterpret_cast<const unsigned char *>(s_image) + threadIdx.x * kPngLdsStride, png_chunk_bytes(P, g % P.chunks_per_row), c);
    if (first) mine += s_codes[2] >> 24;
    if (c.Z != ~0ull) {
#pragma unroll
      for (unsigned w = 0; w < 16u; ++w) {
        const unsigned x = c.d[w];
        unsigned nz = 0x80808080u & ~png_zero_flags(x); /* one iteration per NON-ZERO byte: sparse after the Up filter */
        while (nz) {
          const unsigned sh = (unsigned)__ffs((int)nz) - 8u;
          mine += s_codes[(x >> sh) & 0xffu] >> 24;
          nz &= nz - 1u;
        }
      }
    }
    const unsigned len0 = s_codes[0] >> 24;
    png_for_runs(c.Z, [&](unsigned s, unsigned L) {
      if (L <= 3u) {
        ZL |= ((1ull << L) - 1ull) << s;
        mine += L * len0;
      } else {
        ZL |= 1ull << s;
        MS |= 1ull << s;
        mine += len0 + (s_codes[256u + L - 1u] >> 24);
      }
    });
    if (last) mine += s_codes[256u] >> 24; /* end of block: stored in the unused slot "match of length 0" */
  }
  /* exclusive scan over the workgroup: inside the wave by shuffles, across the four waves through LDS */
  unsigned incl = mine;
  const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  for (int off = 1; off < 64; off <<= 1) {
    const unsigned t = __shfl_up(incl, off, 64);
    if ((int)lane >= off) incl += t;
  }
  if (lane == 63u) s_wave[wave] = incl;
  __syncthreads(); /* every thread has its chunk in registers: the staging buffer becomes the stream image */
  unsigned before = 0, total = 0;
  for (unsigned w = 0; w < kPngBlock / 64; ++w) {
    if (w < wave) before += s_wave[w];
    total += s_wave[w];
  }
  const unsigned long long base = (unsigned long long)P.start_bit[frame] + P.block_bits[(size_t)frame * P.blocks_per_frame + block];
  const unsigned shift = (unsigned)(base & 31ull);
  const unsigned words = (shift + total + 31u) >> 5; /* <= 6241: always inside the image */
  unsigned *dst = P.out + (size_t)frame * P.out_words + (size_t)(base >> 5);
  for (unsigned k = threadIdx.x; k < words; k += kPngBlock) s_image[k] = 0u;
  __syncthreads();
  if (live) {
    const unsigned pos = shift + before + (incl - mine);
    PngEmitSink sink{s_codes, s_image, pos >> 5, pos & 31u, 0ull};
    if (first) sink.put(s_codes[2]);
    const unsigned lo_l = (unsigned)ZL, hi_l = (unsigned)(ZL >> 32);
#pragma unroll
    for (unsigned w = 0; w < 16u; ++w) {
      const unsigned x = c.d[w];
      const unsigned zl = ((w < 8u ? lo_l : hi_l) >> (4u * (w & 7u))) & 0xfu; /* literal zeros among the four bytes */
      /* the word's tokens, one iteration each: its non-zero bytes and its literal zeros (zeros inside a match: none) */
      unsigned tok = png_flags_nibble(0x80808080u & ~png_zero_flags(x)) | zl;
      while (tok) {
        const unsigned b = (unsigned)__ffs((int)tok) - 1u;
        tok &= tok - 1u;
        const unsigned v = (x >> (8u * b)) & 0xffu;
        sink.put(s_codes[v]);
        const unsigned i = 4u * w + b;
        if (v == 0u && ((MS >> i) & 1ull)) {
          const unsigned L = (i == 0u && c.Z == ~0ull) ? 64u : png_run_length(c.Z, i);
          sink.put(s_codes[256u + L - 1u]);
        }
      }
    }
    if (last) sink.put(s_codes[256u]);
    if (sink.fill) atomicOr(&s_image[sink.w], (unsigned)sink.acc);
  }
  __syncthreads();
  for (unsigned k = threadIdx.x; k < words; k += kPngBlock) {
    const unsigned v = s_image[k];
    /* the first word of the span was cleared by png_offsets_kernel and may hold the previous span's tail; the last word is the
     * NEXT span's (cleared) first word -- unless this span ends exactly on a word boundary, in which case it is this
     * workgroup's alone like every word strictly inside, and nobody has cleared it: stored, whatever it holds */
    if (k == 0u || (k + 1u == words && ((shift + total) & 31u) != 0u)) {
      if (v) atomicOr(&dst[k], v);
    } else {
      dst[k] = v;
    }
  }
}

/* CRC-32 of the PNG chunk (type "IDAT" + the frame's finished stream) on the device: the last per-frame host cost of the
 * front end that grew with the stream (0.31 ms of a writer thread per 1080p frame, as much as writing the file).
 * A CRC is linear over GF(2): with raw(B) the register after the bytes B from a zero start,
 *     state(A || B) = state(A) * x^(8 |B|)  XOR  raw(B)      (mod p, reflected arithmetic as in zlib's crc32_combine)
 * so every thread takes 64 bytes of the stream, computes their raw CRC with slice-by-4 tables in LDS, multiplies it by
 * x^(8 x bytes that follow it) (square-and-multiply over a table of x^(2^k)) and XORs the result into the frame's word; the
 * thread holding byte 0 adds the state after "IDAT" moved over the whole stream.  The host appends the four Adler-32 bytes
 * and continues the CRC over them. */
constexpr unsigned kPngCrcPoly = 0xEDB88320u;
constexpr unsigned kPngCrcAfterIdat = 0xCA50F9E1u; /* register after "IDAT" from the all-ones start (tests compare with zlib) */
__device__ __forceinline__ unsigned png_multmodp(unsigned a, unsigned b) { /* a(x) b(x) mod p, bit 31 = x^0 */
  unsigned p = 0u;
#pragma unroll 4
  for (int i = 0; i < 32; ++i) {
    p ^= (a & 0x80000000u) ? b : 0u;
    a <<= 1;
    b = (b >> 1) ^ ((b & 1u) ? kPngCrcPoly : 0u);
  }
  return p;
}
/* x^(8 n) mod p for n < 2^32 from a table of x^(8 d 16^i), d = 0..15, i = 0..7: one multiplication per non-zero hex digit of
 * n (square-and-multiply over x^(2^k) took one per set BIT: 3 x the length of the dependent chain every thread waits for) */
__device__ __forceinline__ unsigned png_x8n(const unsigned *xw, unsigned long long n) {
  unsigned p = 0x80000000u; /* x^0 */
#pragma unroll 1
  for (unsigned i = 0; n; ++i, n >>= 4) {
    const unsigned d = (unsigned)n & 15u;
    if (d) p = png_multmodp(xw[16u * i + d], p);
  }
  return p;
}
__global__ __launch_bounds__(kPngBlock) void png_crc_kernel(const PngParams P) {
  __shared__ unsigned s_t[4][256];
  __shared__ unsigned s_x2n[128];
  __shared__ unsigned s_acc;
  const unsigned frame = blockIdx.y;
  const unsigned long long L = (P.frame_bits[frame] + 7ull) >> 3; /* bytes of the stream (without the Adler-32 trailer) */
  const unsigned long long base = (unsigned long long)blockIdx.x * (kPngBlock * 64u);
  if (base >= L) return; /* the whole workgroup */
  for (unsigned k = threadIdx.x; k < 1024u; k += kPngBlock) (&s_t[0][0])[k] = P.crc_tables[k];
  if (threadIdx.x < 128u) s_x2n[threadIdx.x] = P.crc_tables[1024u + threadIdx.x];
  if (threadIdx.x == 0u) s_acc = 0u;
  __syncthreads();
  const unsigned long long pos = base + (unsigned long long)threadIdx.x * 64u;
  unsigned v = 0u;
  if (pos < L) {
    const unsigned n = (unsigned)min(64ull, L - pos);
    const uint4 *src = reinterpret_cast<const uint4 *>(reinterpret_cast<const unsigned char *>(P.out + (size_t)frame * P.out_words) + pos);
    unsigned d[16];
#pragma unroll
    for (unsigned q = 0; q < 4u; ++q) {
      uint4 t = make_uint4(0u, 0u, 0u, 0u);
      if (16u * q < n) t = src[q]; /* inside the frame's buffer: out_words is rounded up generously */
      d[4 * q + 0] = t.x, d[4 * q + 1] = t.y, d[4 * q + 2] = t.z, d[4 * q + 3] = t.w;
    }
    unsigned c = 0u;
#pragma unroll
    for (unsigned w = 0; w < 16u; ++w) {
      if (4u * w + 4u <= n) {
        c ^= d[w];
        c = s_t[3][c & 0xffu] ^ s_t[2][(c >> 8) & 0xffu] ^ s_t[1][(c >> 16) & 0xffu] ^ s_t[0][c >> 24];
      } else if (4u * w < n) {
        for (unsigned b = 0; b < n - 4u * w; ++b) c = s_t[0][(c ^ (d[w] >> (8u * b))) & 0xffu] ^ (c >> 8);
      }
    }
    const unsigned long long after = L - pos - n;
    v = after ? png_multmodp(png_x8n(s_x2n, after), c) : c;
    if (pos == 0ull) v ^= png_multmodp(png_x8n(s_x2n, L), kPngCrcAfterIdat);
  }
  for (int o = 32; o > 0; o >>= 1) v ^= __shfl_xor(v, o, 64);
  if ((threadIdx.x & 63u) == 0u) atomicXor(&s_acc, v);
  __syncthreads();
  if (threadIdx.x == 0u) atomicXor(&P.crc[frame], s_acc);
}

}  // namespace
