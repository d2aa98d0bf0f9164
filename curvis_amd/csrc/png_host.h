/* png_host.h -- host side of the device PNG front end (kernels_png.h): scratch layout, the launches around the host's
 * Huffman-code construction, assembly of one zlib stream per frame in the caller's buffer.
 * Part of the ONE translation unit curvis_hip.hip (included there, nowhere else). */
#pragma once

namespace {

struct PngScratch { /* one allocation, carved */
  size_t hist, adler, crc, codes, block_bits, start_bit, frame_bits, block_hist, sym_bits, header, crc_tables, out, total;
  size_t out_words;
};

PngScratch png_scratch_layout(const PngParams &P) {
  auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
  PngScratch L{};
  const size_t n = (size_t)P.H * (P.row_bytes + 1);
  /* zlib header (16 bits) + block header (1222) + 12 bits per filtered byte + end of block, rounded up generously */
  L.out_words = (((n * kPngCodeBits + 16 + 1222 + 64) / 32 + 8) + 3) & ~(size_t)3;
  size_t off = 0;
  L.hist = off;
  off = up(off + (size_t)P.n_frames * kPngBins * sizeof(unsigned));
  L.adler = off;
  off = up(off + (size_t)P.n_frames * 2 * sizeof(unsigned long long));
  L.crc = off; /* cleared with the histograms */
  off = up(off + (size_t)P.n_frames * sizeof(unsigned));
  L.codes = off;
  off = up(off + (size_t)P.n_frames * kPngCodes * sizeof(unsigned));
  L.block_bits = off;
  off = up(off + (size_t)P.n_frames * P.blocks_per_frame * sizeof(unsigned long long));
  L.start_bit = off;
  off = up(off + (size_t)P.n_frames * sizeof(unsigned));
  L.frame_bits = off;
  off = up(off + (size_t)P.n_frames * sizeof(unsigned long long));
  L.block_hist = off; /* token counts per workgroup */
  off = up(off + (size_t)P.n_frames * P.blocks_per_frame * kPngBins * sizeof(unsigned short));
  L.sym_bits = off;
  off = up(off + (size_t)P.n_frames * kPngBins * sizeof(unsigned));
  L.header = off;
  off = up(off + (size_t)P.n_frames * kPngHeaderWords * sizeof(unsigned));
  L.crc_tables = off;
  off = up(off + (1024 + 128) * sizeof(unsigned));
  L.out = off;
  off = up(off + (size_t)P.n_frames * L.out_words * sizeof(unsigned));
  L.total = off;
  return L;
}

/* CRC-32 tables for png_crc_kernel: slice-by-4 (reflected 0xEDB88320) and x^(8 d 16^i) mod p for the hex digits d of a byte count */
const std::array<unsigned, 1024 + 128> &png_crc_tables() {
  static const std::array<unsigned, 1024 + 128> T = [] {
    std::array<unsigned, 1024 + 128> t{};
    for (unsigned i = 0; i < 256; ++i) {
      unsigned c = i;
      for (int k = 0; k < 8; ++k) c = (c >> 1) ^ ((c & 1u) ? 0xEDB88320u : 0u);
      t[i] = c;
    }
    for (unsigned s = 1; s < 4; ++s)
      for (unsigned i = 0; i < 256; ++i) t[256 * s + i] = (t[256 * (s - 1) + i] >> 8) ^ t[t[256 * (s - 1) + i] & 0xffu];
    auto mul = [](unsigned a, unsigned b) {
      unsigned p = 0;
      for (int i = 0; i < 32; ++i) {
        if (a & 0x80000000u) p ^= b;
        a <<= 1;
        b = (b >> 1) ^ ((b & 1u) ? 0xEDB88320u : 0u);
      }
      return p;
    };
    unsigned base = 1u << 30; /* x^1 */
    for (int k = 0; k < 3; ++k) base = mul(base, base); /* x^8: one byte */
    for (unsigned i = 0; i < 8; ++i) { /* base = x^(8 16^i) */
      unsigned p = 0x80000000u; /* x^0 */
      for (unsigned d = 0; d < 16; ++d) {
        t[1024 + 16 * i + d] = p;
        p = mul(p, base);
      }
      base = p; /* x^(8 16^(i+1)) */
    }
    return t;
  }();
  return T;
}

/* frames [0, n_frames) of W x H RGB8 in ctx->d_fb -> zlib streams, back to back in `out`; offsets[f] .. offsets[f + 1] is
 * frame f's stream.  kernel_ms: HIP-event time of the five launches (histogram; workgroup bits, offsets, emit, CRC), the
 * host's code construction between the first and the rest excluded. */
int deflate_frames_impl(curvis_ctx *ctx, uint32_t W, uint32_t H, uint32_t n_frames, uint8_t *out, size_t out_cap, size_t *offsets,
                        double *kernel_ms, uint32_t *idat_crc = nullptr, int *crc_valid = nullptr) {
  if (crc_valid) *crc_valid = 0;
  if (!ctx || !out || !offsets || W == 0 || H == 0 || n_frames == 0) return fail(ctx, CURVIS_E_INVALID, "bad argument");
  const size_t frame_bytes = (size_t)W * 3 * H;
  if (!ctx->d_fb || frame_bytes * n_frames > ctx->fb_bytes)
    return fail(ctx, CURVIS_E_INVALID, "the context's framebuffer does not hold that many frames of that size (render first)");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (ctx->streams_pending) { /* the last call's streams are still being read out of the scratch this call is about to overwrite */
    const int rcw = download_wait(ctx);
    if (rcw) return rcw;
  }
  PngParams P{};
  P.fb = ctx->d_fb;
  P.frame_bytes = frame_bytes;
  P.W = W;
  P.H = H;
  P.row_bytes = W * 3;
  P.chunks_per_row = (P.row_bytes + kPngChunk - 1) / kPngChunk;
  const uint64_t cpf = (uint64_t)P.chunks_per_row * H;
  if (cpf > 0x7fffffffull) return fail(ctx, CURVIS_E_INVALID, "frame too large for the device PNG front end");
  P.chunks_per_frame = (unsigned)cpf;
  P.blocks_per_frame = (P.chunks_per_frame + kPngBlock - 1) / kPngBlock;
  P.n_frames = n_frames;
  P.aligned = (P.row_bytes % 4u) == 0u ? 1 : 0;
  P.staged = (P.row_bytes % kPngChunk) == 0u ? 1 : 0;
  P.grid_x = 8u * ((P.blocks_per_frame + 7u) / 8u);
  const PngScratch L = png_scratch_layout(P);
  int rc = ensure_device(ctx, ctx->d_png, ctx->png_cap, L.total);
  if (rc) return rc;
  unsigned char *base = ctx->d_png;
  P.hist = (unsigned *)(base + L.hist);
  P.adler = (unsigned long long *)(base + L.adler);
  P.codes = (const unsigned *)(base + L.codes);
  P.block_bits = (unsigned long long *)(base + L.block_bits);
  P.start_bit = (const unsigned *)(base + L.start_bit);
  P.frame_bits = (unsigned long long *)(base + L.frame_bits);
  P.block_hist = (unsigned short *)(base + L.block_hist);
  P.sym_bits = (const unsigned *)(base + L.sym_bits);
  P.header = (const unsigned *)(base + L.header);
  P.crc_tables = (const unsigned *)(base + L.crc_tables);
  P.crc = (unsigned *)(base + L.crc);
  P.out = (unsigned *)(base + L.out);
  P.out_words = L.out_words;
  const dim3 grid(P.grid_x, n_frames), block(kPngBlock); /* the 8 XCDs take contiguous eighths of a frame: png_logical_block */
  float ms_a = 0.f, ms_b = 0.f;
  const bool dbg_timing = getenv("CURVIS_DEBUG_TIMING") != nullptr;
  auto tnow = [] { return std::chrono::steady_clock::now(); };
  auto tms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
  const auto t_0 = tnow();

  /* pass 1: histograms + Adler sums */
  HIP_TRY(ctx, hipMemsetAsync(base + L.hist, 0, L.codes - L.hist, ctx->stream)); /* hist, adler and the CRC words are adjacent */
  HIP_TRY(ctx, hipEventRecord(ctx->ev0, ctx->stream));
  hipLaunchKernelGGL(png_hist2_kernel, grid, block, 0, ctx->stream, P);
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipEventRecord(ctx->ev1, ctx->stream));
  std::vector<unsigned> hist((size_t)n_frames * kPngBins);
  std::vector<unsigned long long> adler((size_t)n_frames * 2);
  HIP_TRY(ctx, hipMemcpyAsync(hist.data(), base + L.hist, hist.size() * sizeof(unsigned), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(adler.data(), base + L.adler, adler.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  HIP_TRY(ctx, hipEventElapsedTime(&ms_a, ctx->ev0, ctx->ev1));
  const auto t_1 = tnow();

  /* the frames' codes: lengths <= 12 bits over the 286 literal/length symbols (every symbol keeps a code: +1 on each count,
   * which also makes the block header the host writer's), one distance code */
  std::vector<unsigned> codes((size_t)n_frames * kPngCodes, 0u), start_bit(n_frames), sym_bits((size_t)n_frames * kPngBins, 0u);
  std::vector<std::array<uint8_t, 176>> header(n_frames); /* zlib header + block header: 16 + 1222 bits, + BitWriter slack */
  auto build_codes = [&](uint32_t f) {
    uint32_t freq[286];
    for (int i = 0; i < 286; ++i) freq[i] = hist[(size_t)f * kPngBins + i] + 1u;
    freq[256] += 1u; /* the end-of-block symbol */
    uint8_t ll_len[286];
    uint32_t ll[286];
    pngio::huffman_lengths(freq, 286, (int)kPngCodeBits, ll_len);
    pngio::canonical_codes(ll_len, 286, ll);
    unsigned *c = codes.data() + (size_t)f * kPngCodes;
    for (int v = 0; v < 256; ++v) c[v] = (ll[v] & 0xffffu) | ((ll[v] >> 16) << 24);
    c[256] = (ll[256] & 0xffffu) | ((ll[256] >> 16) << 24); /* end of block, in the slot of the impossible "match of length 0" */
    for (int len = 3; len < (int)kPngChunk; ++len) {
      int sym, eb, ev;
      pngio::length_symbol(len, sym, eb, ev);
      const unsigned cl = ll[sym] >> 16;
      const unsigned bits = (ll[sym] & 0xffffu) | ((unsigned)ev << cl); /* + the distance code: one zero bit */
      c[256 + len] = bits | ((cl + (unsigned)eb + 1u) << 24);
      sym_bits[(size_t)f * kPngBins + (size_t)sym] = cl + (unsigned)eb + 1u; /* a match of this symbol: code + extra bits + distance code */
    }
    for (int v = 0; v <= 256; ++v) sym_bits[(size_t)f * kPngBins + (size_t)v] = ll[v] >> 16; /* literals and end of block: the code */
    header[f].fill(0);
    header[f][0] = 0x78;
    header[f][1] = 0x01;
    pngio::BitWriter bw(header[f].data() + 2);
    pngio::put_dynamic_block_header(bw, ll_len);
    start_bit[f] = (unsigned)((bw.p - header[f].data()) * 8 + bw.nb);
    bw.finish();
  };
  /* 286-symbol length-limited Huffman codes, 60-80 us per frame on a host core: nothing next to a render call of a few frames, but a
   * 128-frame call of the efficient renderer spent more time here (9.4 ms, the GPU idle) than in its kernels (profiles/
   * round6_eff_device_sampler.txt) -- the frames are independent, so a large call spreads them over a few threads */
  {
    const unsigned T = n_frames >= 16 ? std::min<unsigned>(8u, n_frames / 8u) : 1u;
    if (T <= 1u) {
      for (uint32_t f = 0; f < n_frames; ++f) build_codes(f);
    } else {
      std::vector<std::thread> th;
      std::atomic<int> failed{0};
      auto work = [&](unsigned t) {
        try {
          for (uint32_t f = t; f < n_frames; f += T) build_codes(f);
        } catch (...) {
          failed = 1;
        }
      };
      try {
        for (unsigned t = 1; t < T; ++t) th.emplace_back(work, t);
      } catch (const std::system_error &) { /* fewer threads than asked for: thread 0 below picks up what has no worker */
      }
      const unsigned started = (unsigned)th.size() + 1u;
      work(0);
      for (unsigned t = started; t < T; ++t) work(t); /* strides nobody was started for */
      for (auto &x : th) x.join();
      if (failed) return fail(ctx, CURVIS_E_INVALID, "out of memory while building the frames' Huffman codes");
    }
  }
  const auto t_2 = tnow();
  HIP_TRY(ctx, hipMemcpyAsync(base + L.codes, codes.data(), codes.size() * sizeof(unsigned), hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(base + L.start_bit, start_bit.data(), start_bit.size() * sizeof(unsigned), hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(base + L.sym_bits, sym_bits.data(), sym_bits.size() * sizeof(unsigned), hipMemcpyHostToDevice, ctx->stream));
  /* the headers go to the device too: the stream it leaves is then complete (but for the Adler-32 trailer), which is what
   * lets it compute the PNG chunk's CRC-32 as well */
  std::vector<unsigned> header_words((size_t)n_frames * kPngHeaderWords, 0u);
  for (uint32_t f = 0; f < n_frames; ++f) std::memcpy(header_words.data() + (size_t)f * kPngHeaderWords, header[f].data(), sizeof header[f]);
  static_assert(sizeof(std::array<uint8_t, 176>) == kPngHeaderWords * sizeof(unsigned), "header staging");
  HIP_TRY(ctx, hipMemcpyAsync(base + L.header, header_words.data(), header_words.size() * sizeof(unsigned), hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(base + L.crc_tables, png_crc_tables().data(), sizeof(unsigned) * (1024 + 128), hipMemcpyHostToDevice, ctx->stream));

  /* offsets from the workgroups' token counts, then ONE more pass over the pixels, then the chunk's CRC over the stream */
  HIP_TRY(ctx, hipEventRecord(ctx->ev0, ctx->stream));
  hipLaunchKernelGGL(png_blockbits_kernel, dim3((P.blocks_per_frame + 3u) / 4u, n_frames), block, 0, ctx->stream, P);
  hipLaunchKernelGGL(png_offsets_kernel, dim3(n_frames), block, 0, ctx->stream, P);
  hipLaunchKernelGGL(png_emit2_kernel, grid, block, 0, ctx->stream, P);
  hipLaunchKernelGGL(png_crc_kernel, dim3((unsigned)((P.out_words * 4 + kPngBlock * 64 - 1) / (kPngBlock * 64)), n_frames), block, 0, ctx->stream, P);
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipEventRecord(ctx->ev1, ctx->stream));
  std::vector<unsigned long long> frame_bits(n_frames);
  HIP_TRY(ctx, hipMemcpyAsync(frame_bits.data(), base + L.frame_bits, frame_bits.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost,
                              ctx->stream));
  std::vector<unsigned> crc_state(n_frames, 0u);
  HIP_TRY(ctx, hipMemcpyAsync(crc_state.data(), base + L.crc, crc_state.size() * sizeof(unsigned), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  HIP_TRY(ctx, hipEventElapsedTime(&ms_b, ctx->ev0, ctx->ev1));
  const auto t_3 = tnow();

  /* streams to the host: deflate bytes, then the Adler-32 of the filtered scanlines.  Sizes first, copies after: a call that
   * fails for want of room must not leave transfers into the caller's buffer in flight */
  size_t off = 0;
  bool fits_scratch = true;
  for (uint32_t f = 0; f < n_frames; ++f) {
    const size_t bytes = (size_t)((frame_bits[f] + 7) / 8);
    if (bytes > L.out_words * 4) fits_scratch = false; /* a frame that does not compress: its stream never left the kernels whole */
    offsets[f] = off;
    off += bytes + 4;
  }
  offsets[n_frames] = off;
  /* what the streams of this call take (option "last_png_stream_bytes"; 0 = a frame did not compress): a caller whose buffer
   * was too small can come back with a larger one instead of giving up on the device front end */
  ctx->last_png_stream_bytes = fits_scratch ? off : 0;
  if (!fits_scratch || off > out_cap) return fail(ctx, CURVIS_E_INVALID, "output buffer too small for the compressed frames");
  /* option "async_streams": the copies go to the copy stream and this call returns without waiting for them -- everything the host
   * still adds (the Adler-32 trailers behind the streams, the CRC-32 over them) needs the sums, not the bytes; the kernels are done
   * (the stream was synchronised for the sizes above).  The bytes are there after curvis_ctx_download_wait. */
  const bool async_out = ctx->async_streams != 0;
  hipStream_t out_stream = ctx->stream;
  if (async_out) {
    const int rcs = ensure_copy_stream(ctx);
    if (rcs) return rcs;
    out_stream = ctx->copy_stream;
  }
  for (uint32_t f = 0; f < n_frames; ++f)
    HIP_TRY(ctx, hipMemcpyAsync(out + offsets[f], (const uint8_t *)(P.out + (size_t)f * L.out_words), offsets[f + 1] - offsets[f] - 4,
                                hipMemcpyDeviceToHost, out_stream));
  if (async_out) {
    HIP_TRY(ctx, hipEventRecord(ctx->ev_streams, ctx->copy_stream));
    ctx->streams_pending = true;
  } else {
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  }
  const auto t_4 = tnow();
  const unsigned long long n = (unsigned long long)H * ((unsigned long long)W * 3 + 1);
  for (uint32_t f = 0; f < n_frames; ++f) {
    /* (the zlib and block headers are already in the stream: png_offsets_kernel wrote them) */
    const unsigned long long s1 = (1ull + adler[(size_t)f * 2]) % 65521ull, s2 = (n % 65521ull + adler[(size_t)f * 2 + 1]) % 65521ull;
    uint8_t *a = out + offsets[f + 1] - 4;
    a[0] = (uint8_t)(s2 >> 8);
    a[1] = (uint8_t)s2;
    a[2] = (uint8_t)(s1 >> 8);
    a[3] = (uint8_t)s1;
    /* CRC-32 of the PNG chunk: the device's register after "IDAT" + stream, finalised, continued over the four trailer bytes */
    if (idat_crc) idat_crc[f] = (uint32_t)crc32((uLong)(crc_state[f] ^ 0xFFFFFFFFu), a, 4);
  }
  if (crc_valid) *crc_valid = 1;
  if (dbg_timing)
    fprintf(stderr, "[curvis] deflate %u frames (ms): histogram pass + sync %.3f (kernel %.3f), codes on the host %.3f, uploads + emit + sync %.3f (kernels %.3f), "
            "streams to the host (%zu bytes) %.3f, trailers %.3f\n", n_frames, tms(t_0, t_1), ms_a, tms(t_1, t_2), tms(t_2, t_3), ms_b, off, tms(t_3, t_4), tms(t_4, tnow()));
  if (kernel_ms) *kernel_ms = (double)ms_a + (double)ms_b;
  ctx->last_png_ms = (double)ms_a + (double)ms_b;
  return CURVIS_OK;
}

}  // namespace
