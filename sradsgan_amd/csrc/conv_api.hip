// C-ABI entry points of the convolution family (include/sradsgan_hip.h): argument checking and the
// choice between the fast kernels (conv_fast.hip) and the generic implicit-GEMM (conv_igemm.hip).
// The choice depends only on the conv's static shape, so srhip_pack_weight and srhip_conv2d_* agree.
#include "conv_internal.h"
#include "conv_dev.h"

using namespace srhip;

namespace srhip {
// one launch re-packs every registered conv weight of a network (after each optimiser step):
// blockIdx.y = entry, blockIdx.x strides over the packed elements
struct PackEntry {
  const float* w;
  float* packed;
  int cout, cin, kh, kw, mode, fast;
};
// Fast entries, round 5: a block walks tiles of RO weight rows (co) x CI source channels (ci) x all taps -- 8 x 64 for the forward
// operand, 32 x 16 for the data-gradient operand (whose 8-channel groups run along co).  The tile is read as it lies (contiguous pieces
// of w[co][ci0 ..][.], coalesced) into LDS and written in two passes of 16-byte stores, each in the order of ITS sections:
//   A: a thread = one 8-channel group of one (destination channel, tap): 32 bytes of the fp32, the split-bf16 (8 hi | 8 lo) and the
//      fp16 (8 | zeros) section; consecutive threads walk the channels of a row: runs of 128 - 256 bytes;
//   B: a thread = one 64-byte row (16-channel chunk, tap, destination channel) of the tiled section, its four quads swizzled;
//      consecutive threads walk the destination channels: runs of 512 - 1024 bytes.
// (conv_internal.h: fast_pack_store describes the layout element by element; this kernel writes the same bytes.  The element-wise
// form read w with a 36-byte stride and stored 2-byte scalars: 623 MB read + 542 MB written per step for 63 MB of weights, 407 us on
// the step's serial tail, profiles/r05_step_traffic.txt.)
__global__ __launch_bounds__(256) void pack_batched_kernel(const PackEntry* __restrict__ tab) {
  const PackEntry e = tab[blockIdx.y];
  const int khkw = e.kh * e.kw;
  if (e.fast) {
    constexpr int LDS_F = 32 * (16 * 9 + 1);                     // >= 8 * (64 * 9 + 1)
    __shared__ float tile[LDS_F];
    const long total = (long)e.cout * e.cin * khkw;
    const int RO = e.mode == 0 ? 8 : 32;
    const int CI = e.mode == 0 ? (khkw <= 9 ? 64 : 16) : (khkw <= 9 ? 16 : 4);
    const int ld = CI * khkw + 1;                                // (+1: rows on different banks)
    const int nci = (e.cin + CI - 1) / CI, nco = (e.cout + RO - 1) / RO;
    const int ndst = e.mode == 0 ? e.cout : e.cin, csrc = e.mode == 0 ? e.cin : e.cout;
    const long ndst16 = ((long)ndst + 15) / 16 * 16;
    float* s0 = e.packed;
    char* s1 = reinterpret_cast<char*>(e.packed + total);
    char* s2 = reinterpret_cast<char*>(e.packed + 2 * total);
    char* s3 = reinterpret_cast<char*>(e.packed + 3 * total);
    for (int t = blockIdx.x; t < nci * nco; t += gridDim.x) {
      const int co0 = (t / nci) * RO, ci0 = (t % nci) * CI;
      const int cw = min(CI, e.cin - ci0), rw = min(RO, e.cout - co0);   // channels / rows of this tile
      const int rowlen = cw * khkw;
      __syncthreads();
      for (int i = threadIdx.x; i < rw * rowlen; i += 256) {
        const int r = i / rowlen, k = i - r * rowlen;
        tile[r * ld + k] = e.w[((size_t)(co0 + r) * e.cin + ci0) * khkw + k];
      }
      __syncthreads();
      // element (destination channel n, tap, source channel c) of the tile: mode 0: n = co0 + a, c = ci0 + b; mode 1: n = ci0 + b, c = co0 + a
      auto at = [&](int a, int b, int tap) { return tile[a * ld + b * khkw + tap]; };
      const int nn = e.mode == 0 ? rw : cw, nc = e.mode == 0 ? cw : rw;  // destination channels / source channels of the tile
      const int n0 = e.mode == 0 ? co0 : ci0, c00 = e.mode == 0 ? ci0 : co0;
      // ---- pass A: row-major sections, source channels fastest
      for (int it = threadIdx.x; it < nn * khkw * (nc / 8); it += 256) {
        const int g8 = it % (nc / 8), rt = it / (nc / 8);
        const int tap = rt % khkw, dn = rt / khkw;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = e.mode == 0 ? at(dn, g8 * 8 + j, tap) : at(g8 * 8 + j, dn, tap);
        const long idx = ((long)(n0 + dn) * khkw + tap) * csrc + c00 + g8 * 8;   // first element of the group (idx % 8 == 0)
        reinterpret_cast<float4*>(s0 + idx)[0] = make_float4(v[0], v[1], v[2], v[3]);
        reinterpret_cast<float4*>(s0 + idx)[1] = make_float4(v[4], v[5], v[6], v[7]);
        bf16x8_t hi, lo;
        split_bf16x8(make_float4(v[0], v[1], v[2], v[3]), make_float4(v[4], v[5], v[6], v[7]), hi, lo);
        *reinterpret_cast<u32x4*>(s1 + idx * 4) = __builtin_bit_cast(u32x4, hi);
        *reinterpret_cast<u32x4*>(s1 + idx * 4 + 16) = __builtin_bit_cast(u32x4, lo);
        const f16x8_t f16 = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3], (_Float16)v[4], (_Float16)v[5], (_Float16)v[6], (_Float16)v[7]};
        const u32x4 z = {0u, 0u, 0u, 0u};
        *reinterpret_cast<u32x4*>(s2 + idx * 4) = __builtin_bit_cast(u32x4, f16);
        *reinterpret_cast<u32x4*>(s2 + idx * 4 + 16) = z;
      }
      // ---- pass B: tiled section, destination channels fastest (c00 % 16 == 0: the tile starts on a chunk boundary)
      for (int it = threadIdx.x; it < nn * khkw * (nc / 16); it += 256) {
        const int dn = it % nn, rt = it / nn;
        const int tap = rt % khkw, g16 = rt / khkw;
        const int n = n0 + dn, swz = (n >> 2) & 3;
        char* row = s3 + (((long)((c00 >> 4) + g16) * khkw + tap) * ndst16 + n) * 64;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          float v[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = e.mode == 0 ? at(dn, g16 * 16 + h * 8 + j, tap) : at(g16 * 16 + h * 8 + j, dn, tap);
          bf16x8_t hi, lo;
          split_bf16x8(make_float4(v[0], v[1], v[2], v[3]), make_float4(v[4], v[5], v[6], v[7]), hi, lo);
          *reinterpret_cast<u32x4*>(row + (((2 * h) ^ swz) << 4)) = __builtin_bit_cast(u32x4, hi);
          *reinterpret_cast<u32x4*>(row + (((2 * h + 1) ^ swz) << 4)) = __builtin_bit_cast(u32x4, lo);
        }
      }
    }
  } else {
    const int csrc = e.mode == 0 ? e.cin : e.cout, cdst = e.mode == 0 ? e.cout : e.cin;
    const int ld = ((cdst + 31) / 32) * 32;
    const long total = (long)khkw * csrc * ld;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
      const int row = (int)(idx / ld), col = (int)(idx - (long)row * ld);
      float v = 0.f;
      if (col < cdst) {
        const int tap = row / csrc, cs = row - tap * csrc;
        const int a = tap / e.kw, b = tap - a * e.kw;
        if (e.mode == 0)
          v = e.w[(((size_t)col * e.cin + cs) * e.kh + a) * e.kw + b];
        else
          v = e.w[(((size_t)cs * e.cin + col) * e.kh + (e.kh - 1 - a)) * e.kw + (e.kw - 1 - b)];
      }
      e.packed[idx] = v;
    }
  }
}
}  // namespace srhip

namespace srhip {
extern int g_fast_cfg;
extern int g_wgrad_cfg;
extern int g_rowtap_addr;
extern int g_rowtap_pipe;
extern int g_patch_ks;
extern int g_fast_dynlds;
extern int g_fast_ablate;
extern int g_conv_math;
extern int g_sgam_cfg;
extern int g_pers_grid;
extern int g_pers_small;
extern int g_pool_epi_any;
extern int g_pers_abl;
extern int g_phase_batch;
extern int g_headconv_rows;
}
extern int g_tail_dbg;
namespace srhip {
}

namespace srhip {
// ---- in-step timing probe (srhip_probe_*): HIP-event pairs around the conv launches of ONE shape, on their launch stream ----
struct Probe {
  int kind = 0, n = 0, h = 0, w = 0, cin = 0, cout = 0, cap = 0, count = 0;
  hipEvent_t ev[2 * 1024];
  int units[1024];
  int created = 0;
};
static Probe g_probe;
static inline bool probe_hit(int kind, int n, int h, int w, int cin, int cout) {
  return g_probe.kind == kind && g_probe.count < g_probe.cap && g_probe.n == n && g_probe.h == h && g_probe.w == w && g_probe.cin == cin &&
         g_probe.cout == cout;
}
static inline int probe_begin(void* stream) {
  const int i = g_probe.count;
  (void)hipEventRecord(g_probe.ev[2 * i], as_stream(stream));
  return i;
}
static inline void probe_end(int i, void* stream, int units = 1) {
  (void)hipEventRecord(g_probe.ev[2 * i + 1], as_stream(stream));
  g_probe.units[i] = units;
  g_probe.count = i + 1;
}
}  // namespace srhip

extern "C" {

/* In-step kernel timing (bench.py: roofline.in_step_*).  srhip_probe_config(kind, ...) arms the probe: every later
 * srhip_conv2d_fwd (kind 1), srhip_conv2d_dgrad (2) or srhip_conv2d_wgrad / _wgrad_multi (3) call with exactly this batch,
 * image and channel geometry records one HIP event before and one after its launches on ITS launch stream, up to `max_pairs`
 * (<= 1024) calls; kind 0 disarms.  srhip_probe_read waits for the recorded events and returns the elapsed milliseconds of each
 * pair (and, in `units`, how many convolutions the call processed: nprob for srhip_conv2d_wgrad_multi, else 1): what the kernel
 * took while the step's other streams shared the chip.  Host-side bookkeeping only; not legal under stream
 * capture; single driving thread. */
int srhip_probe_config(int kind, int n, int h, int w, int cin, int cout, int max_pairs) {
  SRHIP_REQUIRE(kind >= 0 && kind <= 3 && max_pairs >= 0 && max_pairs <= 1024, "probe_config: kind 0..3, max_pairs <= 1024");
  while (g_probe.created < 2 * max_pairs) {
    if (hipEventCreate(&g_probe.ev[g_probe.created]) != hipSuccess) {
      set_error("probe_config: hipEventCreate failed");
      return SRHIP_ERR_LAUNCH;
    }
    ++g_probe.created;
  }
  g_probe.kind = kind; g_probe.n = n; g_probe.h = h; g_probe.w = w; g_probe.cin = cin; g_probe.cout = cout;
  g_probe.cap = kind ? max_pairs : 0;
  g_probe.count = 0;
  return SRHIP_OK;
}
int srhip_probe_read(float* ms, int* units, int cap) {
  SRHIP_REQUIRE(ms || cap == 0, "probe_read: null buffer");
  const int n = g_probe.count < cap ? g_probe.count : cap;
  for (int i = 0; i < n; ++i) {
    if (hipEventSynchronize(g_probe.ev[2 * i + 1]) != hipSuccess || hipEventElapsedTime(&ms[i], g_probe.ev[2 * i], g_probe.ev[2 * i + 1]) != hipSuccess) {
      set_error("probe_read: event %d not readable", i);
      return -1;
    }
    if (units) units[i] = g_probe.units[i];
  }
  return n;
}

/* tuning/experiment knobs; key 0 = fast conv tile configuration (0 = built-in heuristic) */
int srhip_debug_set(int key, int value) {
  if (key == 0) {
    g_fast_cfg = value;
    return SRHIP_OK;
  }
  if (key == 1) {
    g_wgrad_cfg = value;
    return SRHIP_OK;
  }
  if (key == 2) {
    g_fast_dynlds = value;
    return SRHIP_OK;
  }
  if (key == 3) {
    g_fast_ablate = value;
    return SRHIP_OK;
  }
  if (key == 4) {
    g_sgam_cfg = value;
    return SRHIP_OK;
  }
  if (key == 5) {
    g_pers_grid = value;
    return SRHIP_OK;
  }
  if (key == 6) {
    g_pers_abl = value;
    return SRHIP_OK;
  }
  if (key == 7) {
    g_tail_dbg = value;
    return SRHIP_OK;
  }
  if (key == 8) {
    g_rowtap_addr = value;
    return SRHIP_OK;
  }
  if (key == 9) {
    g_rowtap_pipe = value;
    return SRHIP_OK;
  }
  if (key == 10) {
    g_patch_ks = value;
    return SRHIP_OK;
  }
  if (key == 11) {
    g_pers_small = value;
    return SRHIP_OK;
  }
  if (key == 12) {
    g_flat_blocks = value > 0 ? value : 768;
    return SRHIP_OK;
  }
  if (key == 13) {
    g_flat_abl = value;
    return SRHIP_OK;
  }
  if (key == 14) {
    g_flat_f32_k8 = value;
    return SRHIP_OK;
  }
  if (key == 15) {
    g_patch8 = value;
    return SRHIP_OK;
  }
  if (key == 16) {
    g_patch8_abl = value;
    return SRHIP_OK;
  }
  if (key == 17) {
    g_phase_batch = value;
    return SRHIP_OK;
  }
  if (key == 18) {
    g_headconv_rows = value;
    return SRHIP_OK;
  }
  if (key == 19) {
    g_pool_epi_any = value;
    return SRHIP_OK;
  }
  return SRHIP_ERR_ARG;
}

int srhip_set_conv_math(int mode) {
  SRHIP_REQUIRE(mode == SRHIP_MATH_FP32 || mode == SRHIP_MATH_BF16X3 || mode == SRHIP_MATH_HALF, "set_conv_math: unknown mode");
  g_conv_math = mode;
  return SRHIP_OK;
}
int srhip_get_conv_math(void) { return g_conv_math; }

size_t srhip_packed_elems(int cout, int cin, int kh, int kw, int mode) {
  if (cout <= 0 || cin <= 0 || kh <= 0 || kw <= 0 || (mode != 0 && mode != 1)) return 0;
  const bool fast = mode == 0 ? fast_fwd_ok(cin, cout, kh, kw) : fast_dgrad_ok(cin, cout, kh, kw);
  if (fast)   // fp32 + split-bf16 + fp16 sections + the tiled split-bf16 section (fast_pack_store)
    return 3 * (size_t)cout * cin * kh * kw + (size_t)fast_tiled_elems(mode == 0 ? cout : cin, mode == 0 ? cin : cout, kh * kw);
  const int csrc = mode == 0 ? cin : cout, cdst = mode == 0 ? cout : cin;
  return (size_t)kh * kw * csrc * legacy_packed_ld(cdst);
}

int srhip_pack_entry_bytes(void) { return (int)sizeof(PackEntry); }

int srhip_packed_is_fast(int cout, int cin, int kh, int kw, int mode) {
  return (mode == 0 ? fast_fwd_ok(cin, cout, kh, kw) : fast_dgrad_ok(cin, cout, kh, kw)) ? 1 : 0;
}

int srhip_pack_weights_batched(const void* entries_dev, int count, void* stream) {
  SRHIP_REQUIRE(entries_dev && count >= 0, "pack_weights_batched: bad argument");
  if (count == 0) return SRHIP_OK;
  hipLaunchKernelGGL(pack_batched_kernel, dim3(32, count), dim3(256), 0, as_stream(stream),
                     static_cast<const PackEntry*>(entries_dev));
  return check_launch("pack_weights_batched");
}

int srhip_pack_weight(const float* w, float* packed, int cout, int cin, int kh, int kw, int mode, void* stream) {
  SRHIP_REQUIRE(w && packed && cout > 0 && cin > 0 && kh > 0 && kw > 0 && (mode == 0 || mode == 1),
                "pack_weight: bad argument");
  const bool fast = mode == 0 ? fast_fwd_ok(cin, cout, kh, kw) : fast_dgrad_ok(cin, cout, kh, kw);
  if (fast) return fast_pack_weight(w, packed, cout, cin, kh, kw, mode, as_stream(stream));
  return legacy_pack_weight(w, packed, cout, cin, kh, kw, mode, stream);
}

int srhip_conv2d_fwd(const float* x, const float* packed, const float* bias, const float* residual,
                     const float* rowscale, const float* chanscale, float* y, int n, int h, int w, int cin, int cout, int kh, int kw,
                     int stride, int pad, int ldx, int ldy, int ldr, float slope, int flags, void* stream) {
  SRHIP_REQUIRE(x && packed && y, "conv2d_fwd: null tensor");
  SRHIP_REQUIRE(n >= 0 && h > 0 && w > 0 && cin > 0 && cout > 0 && kh > 0 && kw > 0 && stride > 0 && pad >= 0,
                "conv2d_fwd: bad geometry");
  SRHIP_REQUIRE(ldx >= cin && ldy >= cout, "conv2d_fwd: row stride smaller than channel count");
  SRHIP_REQUIRE(!(flags & SRHIP_EPI_BIAS) || bias, "conv2d_fwd: EPI_BIAS without bias");
  SRHIP_REQUIRE(!(flags & SRHIP_EPI_RESIDUAL) || (residual && ldr >= cout), "conv2d_fwd: EPI_RESIDUAL without residual");
  SRHIP_REQUIRE(!(flags & SRHIP_EPI_ROWSCALE) || rowscale, "conv2d_fwd: EPI_ROWSCALE without rowscale");
  SRHIP_REQUIRE(!(flags & SRHIP_EPI_CHANSCALE) || chanscale, "conv2d_fwd: EPI_CHANSCALE without chanscale");
  if (fast_fwd_ok(cin, cout, kh, kw)) {
    const bool pr = probe_hit(1, n, h, w, cin, cout);
    const int pi = pr ? probe_begin(stream) : 0;
    const int rc = fast_conv2d_fwd(x, packed, bias, residual, rowscale, chanscale, y, n, h, w, cin, cout, kh, kw, stride, pad, ldx, ldy,
                                   ldr, slope, flags, as_stream(stream));
    if (pr) probe_end(pi, stream);
    return rc;
  }
  SRHIP_REQUIRE(!(flags & SRHIP_EPI_CHANSCALE), "conv2d_fwd: EPI_CHANSCALE needs Cin % 16 == 0");
  return legacy_conv2d_fwd(x, packed, bias, residual, rowscale, y, n, h, w, cin, cout, kh, kw, stride, pad, ldx, ldy,
                           ldr, slope, flags, stream);
}

/* ABI 9: srhip_conv2d_fwd whose fp32 output y ALSO leaves as padded split-bf16 planes (y_pp, zeroed once by the caller): the
 * attention tail's 1x1 conv (sradsgan.py:262-274) hands the next RAB its input in both forms.  *served = 1 when the kernel that
 * took the launch wrote the planes (the row-group-epilogue kernels of conv_fast.hip), else 0: the caller converts y itself. */
int srhip_conv2d_fwd_dual(const float* x, const float* packed, const float* bias, const float* residual, const float* rowscale,
                          const float* chanscale, float* y, void* y_pp, int* served, int n, int h, int w, int cin, int cout, int kh,
                          int kw, int stride, int pad, int ldx, int ldy, int ldr, float slope, int flags, void* stream) {
  SRHIP_REQUIRE(y_pp && served && (((uintptr_t)y_pp) & 15) == 0 && ldy == cout, "conv2d_fwd_dual: 16-byte aligned plane buffer, dense fp32 output");
  g_dst2_req.pp = y_pp;
  g_dst2_req.served = 0;
  const int rc = srhip_conv2d_fwd(x, packed, bias, residual, rowscale, chanscale, y, n, h, w, cin, cout, kh, kw, stride, pad, ldx, ldy, ldr,
                                  slope, flags, stream);
  *served = g_dst2_req.served;
  g_dst2_req.pp = nullptr;
  g_dst2_req.served = 0;
  return rc;
}

/* ABI 8: a stride-1 3x3 conv to 64 channels (RAB conv2, sradsgan.py:223) that also leaves the CLAM pooling partials of its
 * output (per channel: sum, NaN-propagating maximum, first arg-max pixel; sradsgan.py:108-121) in `pool`: from the conv's own
 * epilogue when the persistent patch kernel takes the launch, else by srhip_clam_pool_partial on y.  *nseg_out = partial segments
 * per image written (the *_pooled tails take it).  pool: three sections of pool_sec_bytes >= n * srhip_clam_pool_max_segments()
 * * 64 * 4 bytes.  flags: 0 or SRHIP_EPI_BIAS.                                                                                  */
int srhip_conv2d_fwd_pool(const float* x, const float* packed, const float* bias, float* y, float* pool, size_t pool_sec_bytes,
                          int* nseg_out, int n, int h, int w, int cin, int cout, int ldx, int ldy, int flags, void* stream) {
  SRHIP_REQUIRE(x && packed && y && pool && nseg_out, "conv2d_fwd_pool: null tensor");
  SRHIP_REQUIRE(cout == 64 && ldy == 64 && n > 0 && h > 0 && w > 0 && cin > 0 && ldx >= cin, "conv2d_fwd_pool: 64 dense destination channels");
  SRHIP_REQUIRE((flags & ~SRHIP_EPI_BIAS) == 0 && (!(flags & SRHIP_EPI_BIAS) || bias), "conv2d_fwd_pool: plain or bias epilogue");
  SRHIP_REQUIRE(pool_sec_bytes % 16 == 0 && pool_sec_bytes < (1u << 30) && (((uintptr_t)pool) & 15) == 0 &&
                    pool_sec_bytes >= (size_t)n * POOL_MAXSEG * 64 * sizeof(float),
                "conv2d_fwd_pool: three 16-byte aligned sections of n * max_segments * 64 floats");
  g_pool_req.out = pool;
  g_pool_req.sec_bytes = (unsigned)pool_sec_bytes;
  g_pool_req.served_nseg = 0;
  const int rc = srhip_conv2d_fwd(x, packed, bias, nullptr, nullptr, nullptr, y, n, h, w, cin, cout, 3, 3, 1, 1, ldx, ldy, 0, 0.f, flags, stream);
  const int served = g_pool_req.served_nseg;
  g_pool_req.out = nullptr;
  g_pool_req.served_nseg = 0;
  if (rc != SRHIP_OK) return rc;
  if (served > 0) {
    *nseg_out = served;
    return SRHIP_OK;
  }
  *nseg_out = srhip_clam_pool_segments();
  return srhip_clam_pool_partial(y, pool, pool_sec_bytes, n, h, w, cout, stream);
}

int srhip_conv2d_dgrad(const float* dy, const float* packed, float* dx, const float* residual, const float* actmask,
                       float slope, int n, int h, int w, int cin, int cout, int kh, int kw, int stride, int pad,
                       int ldy, int ldx, int ldr, int accumulate, void* stream) {
  SRHIP_REQUIRE(dy && packed && dx, "conv2d_dgrad: null tensor");
  SRHIP_REQUIRE(n >= 0 && h > 0 && w > 0 && cin > 0 && cout > 0 && kh > 0 && kw > 0 && stride > 0 && pad >= 0,
                "conv2d_dgrad: bad geometry");
  SRHIP_REQUIRE(pad <= kh - 1 && pad <= kw - 1, "conv2d_dgrad: pad > kernel-1 unsupported");
  SRHIP_REQUIRE(ldy >= cout && ldx >= cin, "conv2d_dgrad: row stride smaller than channel count");
  SRHIP_REQUIRE(!residual || ldr >= cin, "conv2d_dgrad: residual row stride smaller than channel count");
  if (fast_dgrad_ok(cin, cout, kh, kw)) {
    const bool pr = probe_hit(2, n, h, w, cin, cout);
    const int pi = pr ? probe_begin(stream) : 0;
    const int rc = fast_conv2d_dgrad(dy, packed, dx, residual, actmask, slope, n, h, w, cin, cout, kh, kw, stride, pad, ldy,
                                     ldx, ldr, accumulate, as_stream(stream));
    if (pr) probe_end(pi, stream);
    return rc;
  }
  SRHIP_REQUIRE(!residual && !actmask, "conv2d_dgrad: fused residual/activation mask needs Cout % 16 == 0");
  return legacy_conv2d_dgrad(dy, packed, dx, n, h, w, cin, cout, kh, kw, stride, pad, ldy, ldx, accumulate, stream);
}

/* ABI 9: the data gradient with up to THREE residuals, dx = conv_transpose(dy, w) + residual + residual2 + residual3 (in this order):
 * a block input's gradient collects the block's own skip gradient and the gradients of the input's other consumers (the ResGroup's
 * skip, the trunk's bus: sradsgan.py:286-324, 455-460) in the epilogue of the data gradient that is computed last, instead of the
 * two element-wise passes autograd would add.  Contiguous rows only (ldx == ldr == cin).  The persistent patch kernel takes the extra
 * residuals in its epilogue; where another kernel serves the shape / arithmetic mode, one srhip_sum_n pass adds them in the same order. */
static int add_unserved_residuals(float* dx, const float* r2, const float* r3, long count, void* stream) {
  const float* tab[3] = {dx, r2 ? r2 : r3, r3};
  const int k = 1 + (r2 ? 1 : 0) + (r3 ? 1 : 0);
  if (k == 1) return SRHIP_OK;
  return srhip_sum_n(tab, k, dx, count, stream);
}
int srhip_conv2d_dgrad_res3(const float* dy, const float* packed, float* dx, const float* residual, const float* residual2,
                            const float* residual3, int n, int h, int w, int cin, int cout, int kh, int kw, int stride, int pad,
                            void* stream) {
  SRHIP_REQUIRE(residual && (residual2 || residual3), "conv2d_dgrad_res3: residual and at least one more");
  SRHIP_REQUIRE(cin % 4 == 0 && ((((uintptr_t)residual2) | ((uintptr_t)residual3) | ((uintptr_t)dx)) & 15) == 0, "conv2d_dgrad_res3: Cin % 4 == 0, 16-byte aligned tensors");
  g_res_req.r2 = residual2 ? residual2 : residual3;
  g_res_req.r3 = residual2 ? residual3 : nullptr;
  g_res_req.served = 0;
  const int rc = srhip_conv2d_dgrad(dy, packed, dx, residual, nullptr, 0.f, n, h, w, cin, cout, kh, kw, stride, pad, cout, cin, cin, 0, stream);
  const bool served = g_res_req.served != 0;
  g_res_req = ResRequest();
  if (rc != SRHIP_OK || served) return rc;
  return add_unserved_residuals(dx, residual2, residual3, (long)n * h * w * cin, stream);
}
int srhip_conv2d_dgrad_pp_res3(const void* dy, int dy_pp, const float* packed, float* dx, const float* residual, const float* residual2,
                               const float* residual3, int n, int h, int w, int cin, int cout, void* stream) {
  SRHIP_REQUIRE(residual && (residual2 || residual3), "conv2d_dgrad_pp_res3: residual and at least one more");
  SRHIP_REQUIRE(cin % 4 == 0 && ((((uintptr_t)residual2) | ((uintptr_t)residual3)) & 15) == 0, "conv2d_dgrad_pp_res3: Cin % 4 == 0, 16-byte aligned tensors");
  g_res_req.r2 = residual2 ? residual2 : residual3;
  g_res_req.r3 = residual2 ? residual3 : nullptr;
  g_res_req.served = 0;
  const int rc = srhip_conv2d_dgrad_pp(dy, dy_pp, packed, dx, 0, residual, nullptr, 0.f, n, h, w, cin, cout, stream);
  const bool served = g_res_req.served != 0;
  g_res_req = ResRequest();
  if (rc != SRHIP_OK || served) return rc;
  return add_unserved_residuals(dx, residual2, residual3, (long)n * h * w * cin, stream);
}

static size_t align256(size_t b) { return (b + 255) & ~(size_t)255; }

size_t srhip_conv2d_wgrad_workspace(int n, int h, int w, int cin, int cout, int kh, int kw, int stride, int pad) {
  if (n <= 0 || h <= 0 || w <= 0 || cin <= 0 || cout <= 0 || kh <= 0 || kw <= 0 || stride <= 0 || pad < 0) return 0;
  if (fast_wgrad_ok(cin, cout, kh, kw)) return fast_conv2d_wgrad_workspace(n, h, w, cin, cout, kh, kw, stride, pad);
  const int ho = (h + 2 * pad - kh) / stride + 1, wo = (w + 2 * pad - kw) / stride + 1;
  const long rows = (long)n * ho * wo;
  const size_t colsum_b = rows > 0 ? colsum_workspace_bytes(rows, cout) : 0;
  const size_t fused_b = legacy_conv2d_wgrad_bias_workspace(n, h, w, cin, cout, kh, kw, stride, pad);
  return align256(legacy_conv2d_wgrad_workspace(n, h, w, cin, cout, kh, kw, stride, pad)) + (colsum_b > fused_b ? colsum_b : fused_b);
}

int srhip_conv2d_wgrad_can_accumulate(int cin, int cout, int kh, int kw) { return fast_wgrad_ok(cin, cout, kh, kw) ? 1 : 0; }

int srhip_conv2d_wgrad(const float* x, const float* dy, float* dw, float* db, const float* xrowscale,
                       const float* xchanscale, int accumulate, void* workspace,
                       size_t workspace_bytes, int n, int h, int w, int cin, int cout, int kh, int kw, int stride,
                       int pad, int ldx, int ldy, void* stream) {
  SRHIP_REQUIRE(x && dy && dw, "conv2d_wgrad: null tensor");
  SRHIP_REQUIRE(n > 0 && h > 0 && w > 0 && cin > 0 && cout > 0 && kh > 0 && kw > 0 && stride > 0 && pad >= 0,
                "conv2d_wgrad: bad geometry");
  SRHIP_REQUIRE(ldx >= cin && ldy >= cout, "conv2d_wgrad: row stride smaller than channel count");
  if (fast_wgrad_ok(cin, cout, kh, kw)) {
    const bool pr = probe_hit(3, n, h, w, cin, cout);
    const int pi = pr ? probe_begin(stream) : 0;
    const int rc = fast_conv2d_wgrad(x, dy, dw, db, xrowscale, xchanscale, accumulate, workspace, workspace_bytes, n, h, w, cin, cout, kh, kw, stride, pad, ldx,
                                     ldy, as_stream(stream));
    if (pr) probe_end(pi, stream);
    return rc;
  }
  SRHIP_REQUIRE(!xrowscale && !xchanscale && !accumulate,
                "conv2d_wgrad: x scaling / accumulate need Cin % 16 == 0 and Cout % 4 == 0 (srhip_conv2d_wgrad_can_accumulate)");
  const size_t need = srhip_conv2d_wgrad_workspace(n, h, w, cin, cout, kh, kw, stride, pad);
  if (!workspace || workspace_bytes < need) {
    set_error("conv2d_wgrad: workspace %zu bytes < required %zu", workspace_bytes, need);
    return SRHIP_ERR_WORKSPACE;
  }
  const size_t wbytes = align256(legacy_conv2d_wgrad_workspace(n, h, w, cin, cout, kh, kw, stride, pad));
  int bias_done = 0;
  int rc = legacy_conv2d_wgrad(x, dy, dw, workspace, wbytes, n, h, w, cin, cout, kh, kw, stride, pad, ldx, ldy, stream, db,
                               reinterpret_cast<float*>(static_cast<char*>(workspace) + wbytes), &bias_done);
  if (rc || !db || bias_done) return rc;
  const int ho = (h + 2 * pad - kh) / stride + 1, wo = (w + 2 * pad - kw) / stride + 1;
  SRHIP_REQUIRE(cout % 4 == 0 ? cout <= 1024 : cout <= 256, "conv2d_wgrad: too many output channels for the bias sum");
  return colsum_launch(dy, db, static_cast<char*>(workspace) + wbytes, (long)n * ho * wo, cout, ldy, as_stream(stream));
}

/* nprob (2..4) weight gradients of the SAME shape (stride-1 3x3, row-tap eligible, split-bf16 / half arithmetic) in one launch:
 * x[i], dy[i] -> dw[i] (+ db[i] when db and db[i] are non-NULL), written or accumulated into.  See WgradBatch in conv_fast.hip. */
int srhip_conv2d_wgrad_multi_ok(int n, int h, int w, int cin, int cout, int kh, int kw, int stride, int pad) {
  if (n <= 0 || h <= 0 || w <= 0 || !fast_wgrad_ok(cin, cout, kh, kw)) return 0;
  return fast_wgrad_multi_max(n, h, w, cin, cout, kh, kw, stride, pad);
}
int srhip_conv2d_wgrad_multi(int nprob, const float* const* x, const float* const* dy, float* const* dw, float* const* db,
                             int accumulate, void* workspace, size_t workspace_bytes, int n, int h, int w, int cin, int cout,
                             int kh, int kw, int stride, int pad, int ldx, int ldy, void* stream) {
  SRHIP_REQUIRE(x && dy && dw, "conv2d_wgrad_multi: null pointer table");
  SRHIP_REQUIRE(n > 0 && h > 0 && w > 0 && cin > 0 && cout > 0 && ldx >= cin && ldy >= cout, "conv2d_wgrad_multi: bad geometry");
  SRHIP_REQUIRE(srhip_conv2d_wgrad_multi_ok(n, h, w, cin, cout, kh, kw, stride, pad) >= nprob,
                "conv2d_wgrad_multi: shape / size not served for %d problems (srhip_conv2d_wgrad_multi_ok)", nprob);
  const bool pr = probe_hit(3, n, h, w, cin, cout);
  const int pi = pr ? probe_begin(stream) : 0;
  const int rc = fast_conv2d_wgrad_multi(nprob, x, dy, dw, db, accumulate, workspace, workspace_bytes, n, h, w, cin, cout, kh, kw, stride, pad,
                                         ldx, ldy, as_stream(stream));
  if (pr) probe_end(pi, stream, nprob);
  return rc;
}

/* ---- ABI 9: padded split-bf16 planes and the flat weight gradient on them (conv_wgrad_flat.hip) ---- */
int srhip_pp_guard(int w) { return pp_guard(w); }
long srhip_pp_plane_pixels(int n, int h, int w) { return (n > 0 && h > 0 && w > 0) ? pp_plane_pixels(n, h, w) : 0; }
int srhip_pp_from_f32(const float* x, void* pp, int n, int h, int w, int c, int ldx, void* stream) {
  SRHIP_REQUIRE(n > 0 && h > 0 && w > 0 && c > 0 && ldx >= c, "pp_from_f32: bad geometry");
  return pp_from_f32(x, pp, n, h, w, c, ldx, stream);
}
int srhip_pp_to_f32(const void* pp, float* x, int n, int h, int w, int c, int ldx, void* stream) {
  SRHIP_REQUIRE(n > 0 && h > 0 && w > 0 && c > 0 && ldx >= c, "pp_to_f32: bad geometry");
  return pp_to_f32(pp, x, n, h, w, c, ldx, stream);
}
int srhip_conv2d_wgrad_pp_ok(int n, int h, int w, int cin, int cout) {
  if (n <= 0 || h <= 0 || w <= 0 || cin <= 0 || cout <= 0) return 0;
  return flat_wgrad_ok(n, h, w, cin, cout);
}
int srhip_conv2d_pp_ok(int n, int h, int w, int cin, int cout) {
  if (n <= 0 || h <= 0 || w <= 0 || cin <= 0 || cout <= 0 || g_conv_math != 1) return 0;
  if (cin % 32 != 0 || cout % 8 != 0 || cout < 64 || cin < 32 || cout > 512) return 0;      // (> 512: the 4-wave kernel keeps the bias vector in 2 KiB of LDS)
  const long px = pp_plane_pixels(n, h, w);
  return px * (cin > cout ? cin : cout) * 4L < (1L << 31) ? 1 : 0;
}
int srhip_conv2d_fwd_pp(const void* x, int x_pp, const float* packed, const float* bias, void* y, int y_pp, float* pool, size_t pool_sec_bytes,
                        int* nseg_out, int n, int h, int w, int cin, int cout, float slope, int flags, void* stream) {
  SRHIP_REQUIRE(x && packed && y && (x_pp || y_pp), "conv2d_fwd_pp: null tensor / no padded-plane operand");
  SRHIP_REQUIRE(srhip_conv2d_pp_ok(n, h, w, cin, cout), "conv2d_fwd_pp: shape / arithmetic mode not served (srhip_conv2d_pp_ok)");
  SRHIP_REQUIRE((flags & ~(SRHIP_EPI_BIAS | SRHIP_EPI_LRELU)) == 0 && (!(flags & SRHIP_EPI_BIAS) || bias), "conv2d_fwd_pp: bias / LeakyReLU epilogues only");
  SRHIP_REQUIRE((((uintptr_t)x | (uintptr_t)y | (uintptr_t)packed) & 15) == 0, "conv2d_fwd_pp: 16-byte aligned tensors");
  const bool want_pool = pool != nullptr;
  if (want_pool) {
    SRHIP_REQUIRE(nseg_out && !y_pp && cout == 64 && !(flags & SRHIP_EPI_LRELU), "conv2d_fwd_pp: pooling partials: fp32 destination of 64 channels, plain / bias epilogue");
    SRHIP_REQUIRE(pool_sec_bytes % 16 == 0 && pool_sec_bytes < (1u << 30) && (((uintptr_t)pool) & 15) == 0 &&
                      pool_sec_bytes >= (size_t)n * POOL_MAXSEG * 64 * sizeof(float),
                  "conv2d_fwd_pp: three 16-byte aligned sections of n * max_segments * 64 floats");
    g_pool_req.out = pool;
    g_pool_req.sec_bytes = (unsigned)pool_sec_bytes;
    g_pool_req.served_nseg = 0;
  }
  const bool pr = probe_hit(1, n, h, w, cin, cout);
  const int pi = pr ? probe_begin(stream) : 0;
  const int rc = fast_conv2d_fwd_pp(x, x_pp, packed, bias, y, y_pp, n, h, w, cin, cout, cin, cout, slope, flags, as_stream(stream));
  if (pr) probe_end(pi, stream);
  if (want_pool) {
    const int served = g_pool_req.served_nseg;
    g_pool_req.out = nullptr;
    g_pool_req.served_nseg = 0;
    if (rc != SRHIP_OK) return rc;
    if (served > 0) {
      *nseg_out = served;
      return SRHIP_OK;
    }
    *nseg_out = srhip_clam_pool_segments();
    return srhip_clam_pool_partial(static_cast<const float*>(y), pool, pool_sec_bytes, n, h, w, cout, stream);
  }
  return rc;
}
int srhip_conv2d_dgrad_pp(const void* dy, int dy_pp, const float* packed, void* dx, int dx_pp, const float* residual, const void* actmask,
                          float slope, int n, int h, int w, int cin, int cout, void* stream) {
  SRHIP_REQUIRE(dy && packed && dx && (dy_pp || dx_pp), "conv2d_dgrad_pp: null tensor / no padded-plane operand");
  SRHIP_REQUIRE(srhip_conv2d_pp_ok(n, h, w, cout, cin), "conv2d_dgrad_pp: shape / arithmetic mode not served (srhip_conv2d_pp_ok of the transposed conv)");
  SRHIP_REQUIRE(!(residual && dx_pp), "conv2d_dgrad_pp: a residual needs an fp32 destination");
  SRHIP_REQUIRE(!actmask || dx_pp, "conv2d_dgrad_pp: an activation mask (padded planes of the producer's output) needs a padded-plane destination");
  SRHIP_REQUIRE((((uintptr_t)dy | (uintptr_t)dx | (uintptr_t)packed | (uintptr_t)residual | (uintptr_t)actmask) & 15) == 0, "conv2d_dgrad_pp: 16-byte aligned tensors");
  const bool pr = probe_hit(2, n, h, w, cin, cout);
  const int pi = pr ? probe_begin(stream) : 0;
  const int rc = fast_conv2d_dgrad_pp(dy, dy_pp, packed, dx, dx_pp, residual, actmask, slope, n, h, w, cin, cout, cout, cin, cin, as_stream(stream));
  if (pr) probe_end(pi, stream);
  return rc;
}
/* The LeakyReLU mask between a plane-writing forward and the masked data gradient behind it as SIGN WORDS (conv_patch_pers.hip, SIGNS):
 * 1 bit per element in the order the persistent patch kernel's lanes convert them, instead of the consumer re-reading the hi plane of
 * the producer's output (RAB: conv1's LeakyReLU, sradsgan.py:222-223 -- conv2's data gradient reads 3 MB instead of 48 at the bench
 * shape).  The buffer is opaque: only a _dgrad_pp_signs call of the same [n, h, w, channels] geometry may read what _fwd_pp_signs wrote. */
size_t srhip_conv2d_pp_sign_bytes(int n, int h, int w, int cout) {
  if (n <= 0 || h <= 0 || w <= 0 || cout <= 0 || g_conv_math != 1) return 0;
  return (size_t)pp_sign_tiles(n, h, w, cout) * 2048;
}
int srhip_conv2d_fwd_pp_signs(const void* x, int x_pp, const float* packed, const float* bias, void* y_planes, void* signs, size_t sign_bytes,
                              int n, int h, int w, int cin, int cout, float slope, void* stream) {
  SRHIP_REQUIRE(signs && bias && (((uintptr_t)signs) & 15) == 0, "conv2d_fwd_pp_signs: bias and a 16-byte aligned sign buffer");
  const size_t need = srhip_conv2d_pp_sign_bytes(n, h, w, cout);
  SRHIP_REQUIRE(need > 0 && sign_bytes >= need, "conv2d_fwd_pp_signs: shape not served / sign buffer smaller than srhip_conv2d_pp_sign_bytes");
  g_sign_req.words = signs; g_sign_req.bytes = sign_bytes; g_sign_req.mode = 1; g_sign_req.served = 0;
  const int rc = srhip_conv2d_fwd_pp(x, x_pp, packed, bias, y_planes, 1, nullptr, 0, nullptr, n, h, w, cin, cout, slope, SRHIP_EPI_BIAS | SRHIP_EPI_LRELU, stream);
  const bool served = g_sign_req.served != 0;
  g_sign_req = SignRequest();
  if (rc != SRHIP_OK) return rc;
  SRHIP_REQUIRE(served, "conv2d_fwd_pp_signs: the launch that ran does not write sign words");
  return SRHIP_OK;
}
int srhip_conv2d_dgrad_pp_signs(const void* dy, int dy_pp, const float* packed, void* dx_planes, const void* signs, size_t sign_bytes, float slope,
                                int n, int h, int w, int cin, int cout, void* stream) {
  SRHIP_REQUIRE(signs && (((uintptr_t)signs) & 15) == 0, "conv2d_dgrad_pp_signs: a 16-byte aligned sign buffer");
  const size_t need = srhip_conv2d_pp_sign_bytes(n, h, w, cin);
  SRHIP_REQUIRE(need > 0 && sign_bytes >= need, "conv2d_dgrad_pp_signs: shape not served / sign buffer smaller than srhip_conv2d_pp_sign_bytes");
  g_sign_req.words = const_cast<void*>(signs); g_sign_req.bytes = sign_bytes; g_sign_req.mode = 2; g_sign_req.served = 0;
  const int rc = srhip_conv2d_dgrad_pp(dy, dy_pp, packed, dx_planes, 1, nullptr, signs, slope, n, h, w, cin, cout, stream);
  const bool served = g_sign_req.served != 0;
  g_sign_req = SignRequest();
  if (rc != SRHIP_OK) return rc;
  SRHIP_REQUIRE(served, "conv2d_dgrad_pp_signs: the launch that ran does not read sign words");
  return SRHIP_OK;
}
size_t srhip_conv2d_wgrad_pp_workspace(int nprob, int x_pp, int dy_pp, int n, int h, int w, int cin, int cout) {
  if (n <= 0 || h <= 0 || w <= 0 || cin <= 0 || cout <= 0) return 0;
  return flat_wgrad_workspace(nprob, x_pp, dy_pp, n, h, w, cin, cout);
}
int srhip_conv2d_wgrad_pp(int nprob, const void* const* x, const void* const* dy, int x_pp, int dy_pp, float* const* dw, float* const* db,
                          int accumulate, void* workspace, size_t workspace_bytes, int n, int h, int w, int cin, int cout, int ldf,
                          void* stream) {
  SRHIP_REQUIRE(x && dy && dw, "conv2d_wgrad_pp: null pointer table");
  SRHIP_REQUIRE(n > 0 && h > 0 && w > 0 && cin > 0 && cout > 0, "conv2d_wgrad_pp: bad geometry");
  const bool pr = probe_hit(3, n, h, w, cin, cout);
  const int pi = pr ? probe_begin(stream) : 0;
  const int rc = flat_wgrad(nprob, x, dy, x_pp, dy_pp, dw, db, accumulate, workspace, workspace_bytes, n, h, w, cin, cout, ldf, stream);
  if (pr) probe_end(pi, stream, nprob);
  return rc;
}

/* Weight + bias gradient of a conv whose output went through a fused LeakyReLU, from the gradient at the ACTIVATED output:
 * dy_eff = dy * (y > 0 ? 1 : slope) is formed while dy is read (3-channel 3x3 stride-1 head convs at >= 65536 pixels only). */
int srhip_conv2d_wgrad_act(const float* x, const float* dy, const float* y, float slope, float* dw, float* db, void* workspace,
                           size_t workspace_bytes, int n, int h, int w, int cin, int cout, int kh, int kw, int stride,
                           int pad, int ldx, int ldy, void* stream) {
  SRHIP_REQUIRE(x && dy && y && dw && db, "conv2d_wgrad_act: null tensor");
  SRHIP_REQUIRE(!fast_wgrad_ok(cin, cout, kh, kw), "conv2d_wgrad_act: small-channel convs only");
  const size_t need = srhip_conv2d_wgrad_workspace(n, h, w, cin, cout, kh, kw, stride, pad);
  if (!workspace || workspace_bytes < need) {
    set_error("conv2d_wgrad_act: workspace %zu bytes < required %zu", workspace_bytes, need);
    return SRHIP_ERR_WORKSPACE;
  }
  const size_t wbytes = align256(legacy_conv2d_wgrad_workspace(n, h, w, cin, cout, kh, kw, stride, pad));
  int bias_done = 0;
  return legacy_conv2d_wgrad(x, dy, dw, workspace, wbytes, n, h, w, cin, cout, kh, kw, stride, pad, ldx, ldy, stream, db,
                             reinterpret_cast<float*>(static_cast<char*>(workspace) + wbytes), &bias_done, y, slope);
}

int srhip_conv2d_wgrad_act_ok(int n, int h, int w, int cin, int cout, int kh, int kw, int stride, int pad) {
  return (!fast_wgrad_ok(cin, cout, kh, kw) && cin <= 3 && kh == 3 && kw == 3 && stride == 1 && pad == 1 && (long)n * h * w >= 65536 &&
          (size_t)3 * (w + 2) * cin * sizeof(float) <= 32 * 1024) ? 1 : 0;
}

}  // extern "C"
