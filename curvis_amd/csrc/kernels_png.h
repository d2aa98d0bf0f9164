/* kernels_png.h -- PNG front end on the device: the RGB8 frames a render call left in HBM -> one zlib stream per frame.
 * What the reference does with `image::DynamicImage::save` after every frame (src/rendering.rs:110, :311), done where the
 * frame already is, so that 6.2 MB of pixels per 1080p frame neither cross PCIe nor cost a host thread 4-7 ms of
 * filtering and Huffman coding (in `--mode efficient` the HOST was the limit of `curvis video`).
 *
 * Stream format (any PNG decoder reads it): filter type 2 (Up) on every row, ONE dynamic-Huffman deflate block, literals
 * and distance-1 matches for runs of zero bytes -- the format of the host writer (host/png_io.h encode_rgb8_fast) with one
 * difference: a thread tokenises 64 image bytes, so a zero run is cut every 64 bytes (matches of 3..63 instead of 3..258).
 *
 *   png_hist_kernel   pass 1: token histogram (286 symbols) per frame, Adler-32 partial sums of the filtered bytes
 *   (host)            code lengths (<= 12 bits) and canonical codes from the histograms: 286 symbols per frame, microseconds
 *   png_count_kernel  pass 2: bits per workgroup
 *   png_scan_kernel   exclusive scan of the workgroup totals of every frame, total stream length
 *   png_zero_kernel   zeroes exactly the words the stream will occupy
 *   png_emit_kernel   pass 3: every thread writes its codes at its bit offset -- assembled in LDS per workgroup (ds_or), written
 *                     out as whole words (the first and last word of a workgroup's span are shared with its neighbours: atomicOr)
 *
 * This is byte work bound by HBM / L2 traffic, not by FP64 issue: each pass reads the frame (current row + row above) once,
 * pass 3 writes the stream.  Algorithmic bytes per frame: W*H*3 read + stream written.
 * Part of the ONE translation unit curvis_hip.hip (included there, nowhere else). */
#pragma once

namespace {

constexpr unsigned kPngChunk = 64;   /* image bytes per thread */
constexpr unsigned kPngBlock = 256;  /* threads per workgroup */
constexpr unsigned kPngBins = 288;   /* 286 literal/length symbols, padded */
constexpr unsigned kPngCodes = 256 + 64; /* per frame: literal entries [0, 256), match entries for lengths [0, 64) */
constexpr unsigned kPngCodeBits = 12;    /* longest literal/length code */
/* a thread emits at most 65 literals of <= 12 bits (a match replaces >= 3 of them by <= 12 + 3 + 1 bits) = 780 bits */
constexpr unsigned kPngWordsPerThread = 25;
constexpr unsigned kPngLdsWords = kPngBlock * kPngWordsPerThread + 2;

struct PngParams {
  const unsigned char *fb;        /* n_frames frames of H rows of row_bytes bytes, back to back */
  size_t frame_bytes;
  unsigned W, H, row_bytes, chunks_per_row, chunks_per_frame, blocks_per_frame, n_frames;
  int aligned;                    /* row_bytes % 16 == 0: 16-byte loads */
  unsigned *hist;                 /* [n_frames][kPngBins] */
  unsigned long long *adler;      /* [n_frames][2]: sum of the filtered bytes; sum of (n - i) * byte_i; both mod 65521 per workgroup */
  const unsigned *codes;          /* [n_frames][kPngCodes]: bits | n_bits << 24 */
  unsigned long long *block_bits; /* [n_frames][blocks_per_frame]: bits per workgroup, then (in place) exclusive prefix */
  unsigned short *thread_bits;    /* [n_frames][blocks_per_frame * 256]: bits per thread (<= 798), pass 2 -> pass 3 */
  const unsigned *start_bit;      /* [n_frames]: where the token stream starts (after the zlib and the block header) */
  unsigned long long *frame_bits; /* [n_frames]: end of the stream in bits (start offset, tokens, end-of-block code) */
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

/* The tokens of chunk `chunk` of row `row` of frame `frame`, in stream order: lit(value) for a literal byte, match(len) for
 * `len` (3..63) further zero bytes after a literal zero.  Chunk 0 of a row starts with the row's filter-type byte (2 = Up). */
template <typename Lit, typename Match>
__device__ __forceinline__ void png_tokens(const PngParams &P, unsigned frame, unsigned row, unsigned chunk, Lit &&lit, Match &&match) {
  const unsigned x0 = chunk * kPngChunk;
  const unsigned nb = min(kPngChunk, P.row_bytes - x0);
  const unsigned char *cur = P.fb + (size_t)frame * P.frame_bytes + (size_t)row * P.row_bytes + x0;
  const unsigned char *up = cur - P.row_bytes; /* row 0: treated as zeros, never read */
  if (chunk == 0u) lit(2u);
  unsigned run = 0;
  auto flush = [&]() {
    if (run == 0u) return;
    if (run <= 3u) {
      for (unsigned k = 0; k < run; ++k) lit(0u);
    } else {
      lit(0u);
      match(run - 1u);
    }
    run = 0u;
  };
  auto byte = [&](unsigned v) {
    if (v == 0u) {
      ++run;
    } else {
      flush();
      lit(v);
    }
  };
  if (P.aligned) {
    for (unsigned q = 0; q < nb; q += 16u) {
      const uint4 c = *reinterpret_cast<const uint4 *>(cur + q);
      uint4 u = make_uint4(0u, 0u, 0u, 0u);
      if (row != 0u) u = *reinterpret_cast<const uint4 *>(up + q);
      const unsigned d[4] = {png_sub4(c.x, u.x), png_sub4(c.y, u.y), png_sub4(c.z, u.z), png_sub4(c.w, u.w)};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (d[k] == 0u) { /* four zero bytes at once */
          run += 4u;
        } else {
          byte(d[k] & 0xffu);
          byte((d[k] >> 8) & 0xffu);
          byte((d[k] >> 16) & 0xffu);
          byte(d[k] >> 24);
        }
      }
    }
  } else {
    for (unsigned j = 0; j < nb; ++j) byte(((unsigned)cur[j] - (row != 0u ? (unsigned)up[j] : 0u)) & 0xffu);
  }
  flush();
}

__global__ __launch_bounds__(kPngBlock) void png_hist_kernel(const PngParams P) {
  __shared__ unsigned s_hist[kPngBins];
  __shared__ unsigned long long s_sum[2];
  const unsigned frame = blockIdx.y, g = blockIdx.x * kPngBlock + threadIdx.x;
  for (unsigned k = threadIdx.x; k < kPngBins; k += kPngBlock) s_hist[k] = 0u;
  if (threadIdx.x < 2u) s_sum[threadIdx.x] = 0ull;
  __syncthreads();
  if (g < P.chunks_per_frame) {
    const unsigned row = g / P.chunks_per_row, chunk = g - row * P.chunks_per_row;
    const unsigned long long n = (unsigned long long)P.H * (P.row_bytes + 1u);
    /* index of this thread's first filtered byte in the frame's stream (the filter-type byte leads every row) */
    unsigned long long i = (unsigned long long)row * (P.row_bytes + 1u) + (chunk == 0u ? 0u : 1u + chunk * kPngChunk);
    unsigned long long a = 0ull, b = 0ull;
    png_tokens(P, frame, row, chunk,
               [&](unsigned v) {
                 atomicAdd(&s_hist[v], 1u);
                 a += v;
                 b += (n - i) * v; /* Adler-32: s2 = n + sum (n - i) * byte_i */
                 ++i;
               },
               [&](unsigned len) {
                 atomicAdd(&s_hist[png_len_symbol(len)], 1u);
                 i += len;
               });
    atomicAdd(&s_sum[0], a % 65521ull);
    atomicAdd(&s_sum[1], b % 65521ull);
  }
  __syncthreads();
  for (unsigned k = threadIdx.x; k < kPngBins; k += kPngBlock)
    if (s_hist[k]) atomicAdd(&P.hist[(size_t)frame * kPngBins + k], s_hist[k]);
  if (threadIdx.x < 2u) atomicAdd(&P.adler[(size_t)frame * 2 + threadIdx.x], s_sum[threadIdx.x] % 65521ull);
}

/* bits this thread's tokens take with the frame's code (+ the end-of-block code after the last chunk of the frame) */
__device__ __forceinline__ unsigned png_thread_bits(const PngParams &P, const unsigned *s_codes, unsigned frame, unsigned g) {
  unsigned bits = 0;
  if (g < P.chunks_per_frame) {
    const unsigned row = g / P.chunks_per_row, chunk = g - row * P.chunks_per_row;
    png_tokens(P, frame, row, chunk, [&](unsigned v) { bits += s_codes[v] >> 24; }, [&](unsigned len) { bits += s_codes[256u + len] >> 24; });
    if (g == P.chunks_per_frame - 1u) bits += s_codes[256u] >> 24; /* end of block: stored in the unused slot "match of length 0" */
  }
  return bits;
}

__global__ __launch_bounds__(kPngBlock) void png_count_kernel(const PngParams P) {
  __shared__ unsigned s_codes[kPngCodes];
  __shared__ unsigned s_total;
  const unsigned frame = blockIdx.y, g = blockIdx.x * kPngBlock + threadIdx.x;
  for (unsigned k = threadIdx.x; k < kPngCodes; k += kPngBlock) s_codes[k] = P.codes[(size_t)frame * kPngCodes + k];
  if (threadIdx.x == 0u) s_total = 0u;
  __syncthreads();
  unsigned bits = png_thread_bits(P, s_codes, frame, g);
  P.thread_bits[((size_t)frame * P.blocks_per_frame + blockIdx.x) * kPngBlock + threadIdx.x] = (unsigned short)bits;
  for (int off = 32; off > 0; off >>= 1) bits += __shfl_down(bits, off, 64);
  if ((threadIdx.x & 63u) == 0u) atomicAdd(&s_total, bits);
  __syncthreads();
  if (threadIdx.x == 0u) P.block_bits[(size_t)frame * P.blocks_per_frame + blockIdx.x] = s_total;
}

/* one workgroup per frame: block_bits -> exclusive prefix (in place), frame_bits = start + total */
__global__ __launch_bounds__(kPngBlock) void png_scan_kernel(const PngParams P) {
  __shared__ unsigned long long s_part[kPngBlock];
  const unsigned frame = blockIdx.x, nb = P.blocks_per_frame;
  unsigned long long *v = P.block_bits + (size_t)frame * nb;
  const unsigned per = (nb + kPngBlock - 1u) / kPngBlock;
  const unsigned lo = min(nb, threadIdx.x * per), hi = min(nb, lo + per);
  unsigned long long sum = 0ull;
  for (unsigned k = lo; k < hi; ++k) sum += v[k];
  s_part[threadIdx.x] = sum;
  __syncthreads();
  if (threadIdx.x == 0u) { /* 256 partial sums: a serial pass is as fast as anything here */
    unsigned long long run = 0ull;
    for (unsigned k = 0; k < kPngBlock; ++k) {
      const unsigned long long t = s_part[k];
      s_part[k] = run;
      run += t;
    }
    P.frame_bits[frame] = (unsigned long long)P.start_bit[frame] + run;
  }
  __syncthreads();
  unsigned long long run = s_part[threadIdx.x];
  for (unsigned k = lo; k < hi; ++k) {
    const unsigned long long t = v[k];
    v[k] = run;
    run += t;
  }
}

/* zero the words [0, ceil(frame_bits / 32)] of every frame's output (the emit pass ORs into them) */
__global__ __launch_bounds__(kPngBlock) void png_zero_kernel(const PngParams P) {
  const unsigned frame = blockIdx.y;
  const size_t words = min(P.out_words, (size_t)(P.frame_bits[frame] >> 5) + 2u);
  uint4 *dst = reinterpret_cast<uint4 *>(P.out + (size_t)frame * P.out_words);
  const size_t quads = (words + 3u) / 4u; /* out_words is a multiple of 4 */
  for (size_t k = (size_t)blockIdx.x * kPngBlock + threadIdx.x; k < quads; k += (size_t)gridDim.x * kPngBlock) dst[k] = make_uint4(0u, 0u, 0u, 0u);
}

__global__ __launch_bounds__(kPngBlock) void png_emit_kernel(const PngParams P) {
  __shared__ unsigned s_codes[kPngCodes];
  __shared__ unsigned s_wave[kPngBlock / 64];
  __shared__ unsigned s_out[kPngLdsWords];
  const unsigned frame = blockIdx.y, g = blockIdx.x * kPngBlock + threadIdx.x;
  for (unsigned k = threadIdx.x; k < kPngCodes; k += kPngBlock) s_codes[k] = P.codes[(size_t)frame * kPngCodes + k];
  const unsigned mine = P.thread_bits[((size_t)frame * P.blocks_per_frame + blockIdx.x) * kPngBlock + threadIdx.x]; /* counted by pass 2 */
  /* exclusive scan over the workgroup: inside the wave by shuffles, across the four waves through LDS */
  unsigned incl = mine;
  const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  for (int off = 1; off < 64; off <<= 1) {
    const unsigned t = __shfl_up(incl, off, 64);
    if ((int)lane >= off) incl += t;
  }
  if (lane == 63u) s_wave[wave] = incl;
  __syncthreads();
  unsigned before = 0, total = 0;
  for (unsigned w = 0; w < kPngBlock / 64; ++w) {
    if (w < wave) before += s_wave[w];
    total += s_wave[w];
  }
  const unsigned long long base = (unsigned long long)P.start_bit[frame] + P.block_bits[(size_t)frame * P.blocks_per_frame + blockIdx.x];
  const unsigned shift = (unsigned)(base & 31ull);
  const unsigned words = (shift + total + 31u) >> 5; /* what this workgroup's codes occupy: typically a tenth of s_out */
  for (unsigned k = threadIdx.x; k < words; k += kPngBlock) s_out[k] = 0u;
  __syncthreads(); /* also: s_codes complete */
  unsigned pos = shift + before + (incl - mine); /* bit position of this thread's first code in s_out */
  if (g < P.chunks_per_frame) {
    unsigned w = pos >> 5, fill = pos & 31u;
    unsigned long long acc = 0ull;
    auto put = [&](unsigned e) {
      acc |= (unsigned long long)(e & 0xffffffu) << fill;
      fill += e >> 24;
      if (fill >= 32u) {
        atomicOr(&s_out[w], (unsigned)acc);
        ++w;
        acc >>= 32;
        fill -= 32u;
      }
    };
    const unsigned row = g / P.chunks_per_row, chunk = g - row * P.chunks_per_row;
    png_tokens(P, frame, row, chunk, [&](unsigned v) { put(s_codes[v]); }, [&](unsigned len) { put(s_codes[256u + len]); });
    if (g == P.chunks_per_frame - 1u) put(s_codes[256u]);
    if (fill) atomicOr(&s_out[w], (unsigned)acc);
  }
  __syncthreads();
  unsigned *dst = P.out + (size_t)frame * P.out_words + (size_t)(base >> 5);
  for (unsigned k = threadIdx.x; k < words; k += kPngBlock) {
    const unsigned v = s_out[k];
    if (!v) continue;
    if (k == 0u || k + 1u == words)
      atomicOr(&dst[k], v); /* shared with the neighbouring workgroups' spans */
    else
      dst[k] = v;
  }
}

}  // namespace
