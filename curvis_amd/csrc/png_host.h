/* png_host.h -- host side of the device PNG front end (kernels_png.h): scratch
 * layout, the launches (histogram, codes, offsets, emit, CRC: one synchronisation, at the end), assembly of one zlib stream per frame
 * in the caller's buffer.
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
 * frame f's stream.  kernel_ms: HIP-event time of the six launches (histogram, codes, workgroup bits, offsets, emit, CRC). */
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
  P.codes = (unsigned *)(base + L.codes);
  P.block_bits = (unsigned long long *)(base + L.block_bits);
  P.start_bit = (unsigned *)(base + L.start_bit);
  P.frame_bits = (unsigned long long *)(base + L.frame_bits);
  P.block_hist = (unsigned short *)(base + L.block_hist);
  P.sym_bits = (unsigned *)(base + L.sym_bits);
  P.header = (unsigned *)(base + L.header);
  P.crc_tables = (const unsigned *)(base + L.crc_tables);
  P.crc = (unsigned *)(base + L.crc);
  P.out = (unsigned *)(base + L.out);
  P.out_words = L.out_words;
  const dim3 grid(P.grid_x, n_frames), block(kPngBlock); /* the 8 XCDs take contiguous eighths of a frame: png_logical_block */
  float ms_a = 0.f;
  const bool dbg_timing = getenv("CURVIS_DEBUG_TIMING") != nullptr;
  auto tnow = [] { return std::chrono::steady_clock::now(); };
  auto tms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
  const auto t_0 = tnow();

  /* pass 1: histograms + Adler sums; the frames' codes and headers from the histograms (png_codes_kernel: a workgroup per frame --
   * rounds 3-6 built them on the host, between two synchronisations, with the GPU idle for 1.1-1.5 ms of a 128-frame call);
   * offsets from the workgroups' token counts; ONE more pass over the pixels; the chunk's CRC over the stream.  Nothing comes back
   * before the end. */
  HIP_TRY(ctx, hipMemsetAsync(base + L.hist, 0, L.codes - L.hist, ctx->stream)); /* hist, adler and the CRC words are adjacent */
  HIP_TRY(ctx, hipMemcpyAsync(base + L.crc_tables, png_crc_tables().data(), sizeof(unsigned) * (1024 + 128), hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipEventRecord(ctx->ev0, ctx->stream));
  hipLaunchKernelGGL(png_hist2_kernel, grid, block, 0, ctx->stream, P);
  hipLaunchKernelGGL(png_codes_kernel, dim3(n_frames), dim3(kPngCodesThreads), 0, ctx->stream, P);
  hipLaunchKernelGGL(png_blockbits_kernel, dim3((P.blocks_per_frame + 3u) / 4u, n_frames), block, 0, ctx->stream, P);
  hipLaunchKernelGGL(png_offsets_kernel, dim3(n_frames), block, 0, ctx->stream, P);
  hipLaunchKernelGGL(png_emit2_kernel, grid, block, 0, ctx->stream, P);
  hipLaunchKernelGGL(png_crc_kernel, dim3((unsigned)((P.out_words * 4 + kPngBlock * 64 - 1) / (kPngBlock * 64)), n_frames), block, 0, ctx->stream, P);
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipEventRecord(ctx->ev1, ctx->stream));
  std::vector<unsigned long long> adler((size_t)n_frames * 2);
  HIP_TRY(ctx, hipMemcpyAsync(adler.data(), base + L.adler, adler.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
  std::vector<unsigned long long> frame_bits(n_frames);
  HIP_TRY(ctx, hipMemcpyAsync(frame_bits.data(), base + L.frame_bits, frame_bits.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost,
                              ctx->stream));
  std::vector<unsigned> crc_state(n_frames, 0u);
  HIP_TRY(ctx, hipMemcpyAsync(crc_state.data(), base + L.crc, crc_state.size() * sizeof(unsigned), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  HIP_TRY(ctx, hipEventElapsedTime(&ms_a, ctx->ev0, ctx->ev1));
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
    fprintf(stderr, "[curvis] deflate %u frames (ms): launches + sync %.3f (the six kernels %.3f), streams to the host (%zu bytes) %.3f, trailers %.3f\n",
            n_frames, tms(t_0, t_3), ms_a, off, tms(t_3, t_4), tms(t_4, tnow()));
  if (kernel_ms) *kernel_ms = (double)ms_a;
  ctx->last_png_ms = (double)ms_a;
  return CURVIS_OK;
}

}  // namespace
