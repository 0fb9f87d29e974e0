// Windowed multi-head self-attention core of the Swin blocks (BASELINE config 5).
//
// Reference: models/swin_unetr/attention.py:97-120 (WindowAttention.forward) between the two Linear layers:
//   q, k, v = qkv.reshape(b, n, 3, heads, hd).permute(2, 0, 3, 1, 4);  attn = (q * hd^-0.5) @ k^T
//   attn += relative_position_bias[heads, n, n];  attn += mask[window, n, n] (shifted windows, -100 entries,
//   attention.py:123-160);  attn = softmax(attn, -1);  x = (attn @ v).transpose(1, 2).reshape(b, n, C)
// for windows of n = wd*wh*ww tokens (7^3 = 343; 216 at the coarsest level) and head dimension 16 (feature_size 48:
// 48/3 = 96/6 = 192/12 = 384/24).  The qkv and proj Linear layers stay library GEMMs.
//
// One workgroup (4 waves) per (window, head); K and V of the window live in LDS, a wave owns blocks of 32 queries.
// Both products run on MFMA 32x32x16 with fp16 operands and fp32 accumulation, "transposed" so that a query is a LANE:
//   S^T[32 keys][32 queries] = K_blk [32 x 16] . Q^T [16 x 32]           (one instruction per block: K = head dim = 16)
// leaves the whole score row of a query in the registers of two lanes (l, l + 32): bias, mask, the running max, exp and the
// sum are plain register arithmetic + one exchange between the halves per key block -- no shuffles per element.  The accumulator
// tile is then the B operand of the second product as it stands (cdna_hip_programming.md, "An accumulator tile as the
// next MFMA's operand"):
//   O^T[dims][32 queries] += V^T_blk [dims x 32 keys, keys in the accumulator's row order] . P^T_blk [32 keys x 32 queries]
// (rows 16..31 of the 32-row output are padding: the price of a 16-wide head on a 32-row instruction).  V is written
// to LDS already in that permuted key order, so its operand fragment is one 16-byte read.
// The relative-position bias is either dense and TRANSPOSED (bias_t) or -- the production form -- the reference's own
// table: bias[h][q][k] = table[relative_position_index[q][k]][h] (attention.py:103-106), and the index is a function of
// the coordinate difference of q and k inside the table's (2gd-1)(2gh-1)(2gw-1) grid (attention.py:56-73), so the kernel
// keeps the 2197-entry column of its head in LDS (8.8 KB instead of 343 x 343 floats from L2) and looks up
// table[off(q) - off(k) + centre].  Token -> coordinate uses the TABLE's grid (7, 7, 7) even for clipped windows: the
// reference slices relative_position_index[:n, :n] of the 7^3 index (attention.py:104).
// A third form -- the bias gathered on the host side into the accumulator's order, four 16-byte global loads per key block instead
// of the LDS lookups -- measured slower at the coarse stages and was removed in round 4 with its ABI parameter.
// Bias and mask otherwise arrive TRANSPOSED ([head][key][query], [window][key][query], fp32) so that the 32 lanes of a half read
// 128 contiguous bytes per key.  The shifted-window mask can instead be given as what compute_mask builds it from: one
// region id per token of every window ([windows per image][tokens], uint8; attention.py:135-157) -- 343 bytes per window
// instead of 343 x 343 floats (161 MB per image at 48^3 tokens); the kernel adds -100 where the ids of query and key differ.
#include "common.hpp"
#include "../../include/dua_hip.h"

namespace dua {

namespace wa {
constexpr int HD = 16;          // head dimension
constexpr int MAXB = 11;        // 32-token blocks per window (n <= 352)
constexpr int MAXTAB = 2208;    // (2*7-1)^3 = 2197 table entries per head, padded
}  // namespace wa

struct WinAttnArgs {
  const void* qkv; const float* bias_t; const float* mask_t; const unsigned char* region; void* out;
  const float* table;           // [heads][tab_len] or null (then bias_t)
  int gd, gh, gw, tab_len;      // grid the relative-position index was built for
  int n, heads, nw;             // tokens per window, heads, windows per image (mask index = window % nw)
  float scale;
};

// BIAS: where the relative-position bias comes from -- 0 the reference's table (production), 2 a dense transposed matrix (the
// kernel tests' independent form).  A template parameter: as run-time branches the forms left the score tile
// "undefined on the other paths", which hipcc materialises as 16 register fills per key block.
template <typename T, int BIAS>
__global__ __launch_bounds__(256) void window_attention_kernel(WinAttnArgs a) {
  using namespace wa;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  f16* Kl = (f16*)smem;                           // [nb*32][16]
  f16* Vp = Kl + MAXB * 32 * HD;                     // [nb][2 s][2 hh][16 rows (dims)][8 keys]  (permuted V^T; MFMA rows 16..31 are zeros)
  float* tab = (float*)(Vp + MAXB * 2 * 2 * 16 * 8);                // [MAXTAB] bias table column of this head (x log2 e)
  short* koff = (short*)(tab + MAXTAB);                              // [nb*32] coordinate offset of every token
  unsigned char* regl = (unsigned char*)(koff + MAXB * 32);          // [nb*32] region id of every token of the window
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, hh = lane >> 5;
  const int win = blockIdx.x, head = blockIdx.y;
  const int n = a.n, nb = (n + 31) >> 5, C = a.heads * HD;
  const T* base = (const T*)a.qkv + (long)win * n * 3 * C + head * HD;
  // ---- stage K (row major), V (permuted, zero padded) and this workgroup's Q blocks: one 16-byte load per (token, half
  // of the head dimension) and operand; V^T is built by scattering the 8 dims of a key into its permuted column (2-byte LDS
  // writes) -- gathering it with 2-byte GLOBAL loads made the staging the longest phase of the workgroup ----
  auto load8 = [](const T* p) {
    f16x8 o;
    if constexpr (sizeof(T) == 2) {
      o = *(const f16x8*)p;
    } else {
      const f32x4 lo = *(const f32x4*)p, hi = *(const f32x4*)(p + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { o[e] = (f16)lo[e]; o[4 + e] = (f16)hi[e]; }
    }
    return o;
  };
  for (int i = tid; i < nb * 32 * 2; i += 256) {          // (token, half of the 16 dims)
    const int tok = i >> 1, half = i & 1;
    f16x8 k, v;
#pragma unroll
    for (int e = 0; e < 8; ++e) { k[e] = (f16)0.f; v[e] = (f16)0.f; }
    if (tok < n) {
      const T* p = base + (long)tok * 3 * C + half * 8;
      k = load8(p + C);
      v = load8(p + 2 * C);
    }
    *(f16x8*)(Kl + tok * HD + half * 8) = k;
    // Vp[kb][s][h2][row][j] = V[key = kb*32 + 16 s + 8 (j >> 2) + 4 h2 + (j & 3)][dim = row], row < 16
    const int kb = tok >> 5, u = tok & 15, s2 = (tok >> 4) & 1;
    const int h2 = (u >> 2) & 1, jj = ((u >> 3) << 2) | (u & 3);
    f16* vp = Vp + (((kb * 2 + s2) * 2 + h2) * 16 + half * 8) * 8 + jj;
#pragma unroll
    for (int e = 0; e < 8; ++e) vp[e * 8] = v[e];
  }
  constexpr bool has_table = BIAS == 0;
  const int sh_ = 2 * a.gw - 1, sd_ = (2 * a.gh - 1) * sh_;         // strides of the (dd, dh, dw) difference grid
  // The score tile starts as the BIAS: the table lookups land in the accumulator registers and the MFMA adds q.k on top
  // (q carries the scale: head dimension 16 -> 0.25, exact in fp16), so a score costs no zero, no multiply-add of its own;
  // exp(z - m) is exp2(fma(z, log2 e, -m log2 e)).  koff holds BYTE offsets into the table: a lookup address is one subtraction.
  constexpr float LOG2E = 1.4426950408889634f;
  if (has_table) {
    for (int i = tid; i < a.tab_len; i += 256) tab[i] = a.table[(long)head * a.tab_len + i];
    for (int i = tid; i < nb * 32; i += 256) {
      const int t = i < n ? i : 0;
      koff[i] = (short)(4 * ((t / (a.gh * a.gw)) * sd_ + ((t / a.gw) % a.gh) * sh_ + t % a.gw));
    }
  }
  // Shift mask: most windows of a shifted grid lie inside ONE region (only the last window along an axis is assembled from
  // wrapped pieces: 127 of 343 at the 48^3-token stage) -- their mask is all zeros and the compare per score is skipped.
  int mixed = 0;
  if (a.region != nullptr) {
    const unsigned char* rg = a.region + (long)(win % a.nw) * n;
    const unsigned char r0 = rg[0];
    for (int i = tid; i < nb * 32; i += 256) {
      const unsigned char v = i < n ? rg[i] : (unsigned char)255;
      regl[i] = v;
      mixed |= (i < n && v != r0) ? 1 : 0;
    }
  }
  const bool has_region = __syncthreads_or(mixed) != 0;

  const float* bias = BIAS == 2 ? a.bias_t + (long)head * n * n : nullptr;
  const float* mask = a.mask_t ? a.mask_t + (long)(win % a.nw) * n * n : nullptr;
  T* outp = (T*)a.out + (long)win * n * C + head * HD;
  // A wave owns query blocks blockIdx.z * 4 + wave, + 4 * gridDim.z, ...: with one workgroup per (window, head) (gridDim.z = 1,
  // the 48^3-token stage: 1029 workgroups, all resident at five per CU) K and V are staged once for the whole window; the
  // coarse stages split the query blocks over gridDim.z workgroups to have enough of them.
#if defined(WA_ABL) && (WA_ABL & 16)
  if (a.scale != -1.f) return;                   // ablation: staging only
#endif
  for (int qb = blockIdx.z * 4 + wave; qb < nb; qb += 4 * gridDim.z) {
  const int q = qb * 32 + r;                     // this lane's query
  const bool qok = q < n;
  const int qc = qok ? q : 0;
  f16x8 qf;                                      // B operand: Q^T[dims 8hh..][query r], straight from global memory
#pragma unroll
  for (int e = 0; e < 8; ++e) qf[e] = (f16)0.f;
  if (qok) qf = load8(base + (long)q * 3 * C + hh * 8);
#pragma unroll
  for (int e = 0; e < 8; ++e) qf[e] = (f16)((float)qf[e] * a.scale);
  const unsigned char rq = has_region ? regl[qb * 32 + r] : (unsigned char)0;
  // LDS byte address of this query's table entry for a key at offset 0: a lookup address is ONE subtraction (koff[key])
  const unsigned qtab = has_table ? (unsigned)(size_t)(__attribute__((address_space(3))) char*)tab + (unsigned)((int)koff[qc] +
                                    4 * ((a.gd - 1) * sd_ + (a.gh - 1) * sh_ + (a.gw - 1))) : 0u;
  const int nlast = n - (nb - 1) * 32 - 4 * hh;      // last key block: accumulator rows (i & 3) + 8 (i >> 2) >= nlast are padding keys
  f16x8 zero8, ones8;
#pragma unroll
  for (int e = 0; e < 8; ++e) { zero8[e] = (f16)0.f; ones8[e] = (f16)1.f; }
  // score tile of key block kb for this lane's query: scale * <q, k> + bias + mask (padding keys: -3e38)
  auto scores = [&](int kb) {
    const f16x8 kf = *(const f16x8*)(Kl + (kb * 32 + r) * HD + hh * 8);  // A operand: K[key r][dims 8hh..]
    f32x16 z;
    if constexpr (has_table) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {               // register quad j = keys kb*32 + 8j + 4hh + (0..3): one 8-byte read of offsets
        typedef short short4v __attribute__((ext_vector_type(4)));
        const short4v ko = *(const short4v*)(koff + kb * 32 + 8 * j + 4 * hh);
#pragma unroll
#if defined(WA_ABL) && (WA_ABL & 1)
        for (int e = 0; e < 4; ++e) z[4 * j + e] = (float)ko[e] * 1e-9f;      // ablation: no table lookup
#else
        for (int e = 0; e < 4; ++e) z[4 * j + e] = *(const __attribute__((address_space(3))) float*)(size_t)(qtab - (unsigned)(int)ko[e]);
#endif
      }
    } else {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int key = kb * 32 + acc_row(i, hh);
        z[i] = bias[(long)(key < n ? key : 0) * n + qc];
      }
    }
    z = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf, z, 0, 0, 0);
    if (mask) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int key = kb * 32 + acc_row(i, hh);
        z[i] += mask[(long)(key < n ? key : 0) * n + qc];
      }
    }
#if defined(WA_ABL) && (WA_ABL & 4)
    if (false) {
#else
    if (has_region) {                             // compute_mask's 0 / -100 from the region ids
#endif
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const unsigned rk = *(const unsigned*)(regl + kb * 32 + 8 * j + 4 * hh);
#pragma unroll
        for (int e = 0; e < 4; ++e) z[4 * j + e] += ((rk >> (8 * e)) & 255u) != (unsigned)rq ? -100.f : 0.f;
      }
    }
    if (kb == nb - 1) {                           // only the last block holds padding keys: they never win the max
#pragma unroll
      for (int i = 0; i < 16; ++i) z[i] = (i & 3) + 8 * (i >> 2) < nlast ? z[i] : -3.0e38f;
    }
    return z;
  };
  // One pass with a running maximum (the 11 score tiles of a query would not fit the register file next to the output
  // tile, and recomputing them costs as much as the softmax itself): when a block raises the maximum, the sum and the
  // eight live registers of O^T are rescaled by exp(old - new).  Both halves of a query's keys feed the same MFMA
  // contraction, so they share one maximum per block (one exchange with lane ^ 32).
  float mx = -3.0e38f;
  f32x16 O;
#pragma unroll
  for (int i = 0; i < 16; ++i) O[i] = 0.f;
  for (int kb = 0; kb < nb; ++kb) {
    const f32x16 z = scores(kb);
    float bm = z[0];
#pragma unroll
    for (int i = 1; i < 16; ++i) bm = fmaxf(bm, z[i]);
    bm = fmaxf(bm, other_half(bm));
#if defined(WA_ABL) && (WA_ABL & 8)
    const float mnew = 0.f * bm;                  // ablation: no running maximum, no rescale
#else
    const float mnew = fmaxf(mx, bm);
    const float alpha = __builtin_amdgcn_exp2f((mx - mnew) * LOG2E);
    mx = mnew;
#pragma unroll
    for (int i = 0; i < 9; ++i) O[i] *= alpha;
#endif    // dims 0..15 (registers 0..7) and the denominator row 16 (register 8, lanes hh = 0)
    f16x8 p[2];
    const float moff = -mnew * LOG2E;
#pragma unroll
#if defined(WA_ABL) && (WA_ABL & 2)
    for (int i = 0; i < 16; ++i) p[i >> 3][i & 7] = (f16)fminf(z[i] - mnew + 0.f * moff, 1.f);      // ablation: no exp2
#else
    for (int i = 0; i < 16; ++i) p[i >> 3][i & 7] = (f16)__builtin_amdgcn_exp2f(fmaf(z[i], LOG2E, moff));
#endif
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      f16x8 vf = r == HD ? ones8 : zero8;          // rows 17..31 of the A operand are padding; row 16 is all ones: its output
      if (r < HD) vf = *(const f16x8*)(Vp + ((long)((kb * 2 + s) * 2 + hh) * 16 + r) * 8);   // row is the softmax denominator
      O = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, p[s], O, 0, 0, 0);
    }
  }
  const float inv = 1.f / (O[8] + other_half(O[8]));     // row 16 lives in the hh = 0 lanes (row 20, hh = 1, is zero)
  // O^T rows = dims: register i of half hh holds dim (i & 3) + 8 (i >> 2) + 4 hh; dims < 16 are i = 0..7
  if (qok) {
    T* o = outp + (long)q * C;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      typedef T TV4 __attribute__((ext_vector_type(4)));
      TV4 w;
#pragma unroll
      for (int e = 0; e < 4; ++e) w[e] = (T)(O[4 * g + e] * inv);
      *(TV4*)(o + 8 * g + 4 * hh) = w;
    }
  }
  }
}

}  // namespace dua

extern "C" int dua_window_attention_fwd(int dtype, int windows, int tokens, int heads, int windows_per_image,
                                        const void* qkv, const float* bias_t, const float* mask_t,
                                        const unsigned char* region_ids, const float* bias_table, int grid_d, int grid_h,
                                        int grid_w, float scale, void* out, void* stream) {
  using namespace dua;
  if (!qkv || (!bias_t && !bias_table) || !out || windows <= 0 || heads <= 0 || tokens <= 0 || tokens > wa::MAXB * 32)
    return DUA_ERR_ARG;
  if ((mask_t || region_ids) && (windows_per_image <= 0 || windows % windows_per_image)) return DUA_ERR_ARG;
  WinAttnArgs a;
  a.qkv = qkv; a.bias_t = bias_table ? nullptr : bias_t; a.mask_t = mask_t; a.region = region_ids; a.out = out;
  a.table = bias_table; a.gd = grid_d; a.gh = grid_h; a.gw = grid_w; a.tab_len = 0;
  if (bias_table) {
    if (grid_d <= 0 || grid_h <= 0 || grid_w <= 0 || tokens > grid_d * grid_h * grid_w) return DUA_ERR_ARG;
    a.tab_len = (2 * grid_d - 1) * (2 * grid_h - 1) * (2 * grid_w - 1);
    if (a.tab_len > wa::MAXTAB) return DUA_ERR_ARG;
  }
  a.n = tokens; a.heads = heads; a.nw = (mask_t || region_ids) ? windows_per_image : 1; a.scale = scale;
  const int nb = (tokens + 31) / 32;
  const int lds = wa::MAXB * 32 * wa::HD * 2 + wa::MAXB * 2 * 2 * 16 * 8 * 2 +                            // K, V^T
                  wa::MAXTAB * 4 + wa::MAXB * 32 * 2 + wa::MAXB * 32;                                        // table, offsets, regions
  dim3 grid(windows, heads, (long)windows * heads >= 1024 ? 1 : (nb + 3) / 4);
  if (dtype != DUA_F16 && dtype != DUA_F32) return DUA_ERR_ARG;
  const int bias_mode = a.table ? 0 : 2;
  auto go = [&](auto kern) { hipLaunchKernelGGL(kern, grid, dim3(256), lds, (hipStream_t)stream, a); };
  if (dtype == DUA_F16) {
    if (bias_mode == 0) go(window_attention_kernel<f16, 0>); else go(window_attention_kernel<f16, 2>);
  } else {
    if (bias_mode == 0) go(window_attention_kernel<float, 0>); else go(window_attention_kernel<float, 2>);
  }
  return (int)hipGetLastError();
}
