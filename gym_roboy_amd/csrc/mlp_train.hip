// mlp_train.hip - the PPO minibatch gradient of MlpPolicy as one kernel per network on the matrix cores
// (include/roboy_policy.h: rp_ppo_grad_dev; gym_roboy_amd/ppo.py: _minibatch_loss is the torch statement of it).
//
// For a minibatch (obs, act, normalised advantage, old log-probability, old value, return), one launch per net
// (NET 0: action mean + log-std, the clipped surrogate; NET 1: value, the clipped value loss) runs, per 64-sample
// tile of a wave:  forward (as mlp_policy.hip; the activations stay in registers)  ->  per-sample loss derivative
// delta3 (one sample per lane)  ->  delta2 = (W3^T delta3) (1 - h2^2),  delta1 = (W2^T delta2) (1 - h1^2)  with
// the same "a layer's result registers are the next layer's B operands" chaining (the transposed weights are
// packed as A operands by rp_pack_train)  ->  the weight gradients  dW3 += delta3 h2^T, dW2 += delta2 h1^T,
// dW1 += delta1 [obs | 1]^T.  Those contract over SAMPLES, which sit on the lanes of every activation register,
// so both factors go through a 32x32 transpose in LDS (33-float rows: conflict-free both ways) into the layout
// "units on the lanes, samples in the registers"; there register r of the two factors is directly an A / B operand
// pair whose K pair is the samples (U(r), U(r) + 4) of the column tile.  The gradient accumulators (64 + 32 + 32
// registers of 32x32 tiles) live across all tiles of a wave; bias and log-std gradients are per-lane sums reduced
// once at the end.  Each wave writes its partial gradient (torch parameter order); a second kernel sums the waves.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>

#include "../../include/roboy_policy.h"
#include "mlp_common.hpp"

namespace {
using namespace rpd;

__device__ __forceinline__ void wave_fence() {         // orders this wave's LDS traffic (in-order per wave)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// 32x32 transpose of an accumulator tile through the wave's LDS scratch T (32 rows of 33 floats):
// in: lane (column n = l & 31, half h), register r <-> row U(r) + 4 h;  out: the same with rows and columns exchanged
__device__ __forceinline__ f32x16 transpose_tile(const f32x16 &d, float *T, int col, int half) {
#pragma unroll
    for (int r = 0; r < 16; ++r) T[(unit_of(r) + 4 * half) * 33 + col] = d[r];
    wave_fence();
    f32x16 o;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] = T[col * 33 + unit_of(r) + 4 * half];
    wave_fence();
    return o;
}

// One dword per lane from this lane's global address straight into LDS at (wave-uniform byte address) + 4 * lane: no
// register destination, so a tile's inputs can be in flight under the previous tile's arithmetic in a kernel that has no
// registers to spare.  hipcc does not count it: the consumer waits with wait_dma() (vmcnt covers loads in issue order).
__device__ __forceinline__ void dma_dword(const float *gsrc, const float *lds_dst) {
    const unsigned at = __builtin_amdgcn_readfirstlane(unsigned(reinterpret_cast<uintptr_t>(lds_dst)));   // LDS aperture: low 32 bits
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(at) : "memory");
}
__device__ __forceinline__ void wait_dma() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// A value read from a clamped address and then selected (`in range ? value : constant`) is what keeps an MFMA stage free of
// branches - but hipcc sinks such a read back under the condition: an EXEC-masked block per element (s_and_saveexec, address
// arithmetic, ds_read, s_or, s_waitcnt lgkmcnt(0)), i.e. one exposed LDS latency per MFMA pair and no scheduling across them.
// Passing the value through an empty asm statement pins the read where it is written.
__device__ __forceinline__ float pinned(float v) { asm volatile("" : "+v"(v)); return v; }
constexpr int PFS = 65;                                   // row stride of the prefetched inputs: [row][sample], conflict-free both ways

struct TrainArgs {
    const float *packed;                                  // rp_pack_train blob
    const float *obs, *act, *adv, *logp_old, *val_old, *ret;
    const long long *index;                               // row of sample i in obs / act / logp_old / val_old / ret (NULL: i); adv is direct ...
    const float *adv_stats;                               // ... unless this is given: {mean, 1 / (std + 1e-8)} of the minibatch's advantages
                                                          // (rp_adv_stats_dev); adv is then indexed like the rest and normalised here
    float *partials;                                      // [waves][gstride]
    long B;
    int obs_dim, act_dim, gstride;
    float cliprange, vf_coef, inv_B;
};

// KX: 32-column tiles of [obs | 1] (obs_dim + 1 <= 32 KX);  NJ: compile-time bound of the outputs (n_out <= NJ,
// a multiple of 8): the per-sample arrays of the loss derivative are NJ registers each
// PF (KX == 1 only): the tile's rows (observation, action, advantage, old log-probability | old value, return) arrive by
// LDS-DMA one tile ahead, double-buffered per wave; without it they are loaded when the tile starts (the indexed minibatch
// of 8 388 608 samples: 4.4 ms against 3.5 contiguous, 29 % of the wave cycles waiting)
template <int NET, int KX, int NJ, bool PF>
__global__ void __launch_bounds__(256, 1)
mlp_grad_kernel(const TrainArgs a) {
    static_assert(!PF || KX == 1, "the prefetching form is the small instance's");
    constexpr int OT = (NJ + 31) / 32;                     // 32-row tiles of the outputs
    extern __shared__ float4 lds4[];
    float *lds = reinterpret_cast<float *>(lds4);
    const int obs_dim = a.obs_dim, act_dim = a.act_dim, n_out = NET == 0 ? a.act_dim : 1;
    // LDS holds THIS net's operand blocks only (offsets of the blob rebased), then per-wave scratch
    const Layout G = layout_of(obs_dim, act_dim);
    Layout L = G;
    const int ot_net = NET == 0 ? G.ot_pi : 1;
    int lds_used = 0;
    auto stage = [&](int src, int n_floats) {              // blob block -> LDS at lds_used; all blocks are multiples of 4 floats
        const float4 *s4 = reinterpret_cast<const float4 *>(a.packed + src);
        for (int k = threadIdx.x; k < n_floats / 4; k += blockDim.x) lds4[lds_used / 4 + k] = s4[k];
        const int at = lds_used;
        lds_used += n_floats;
        return at;
    };
    L.o_l1 = stage(G.o_l1 + (NET == 0 ? 0 : HT) * G.k1s * 64, HT * G.k1s * 64) - 0;
    L.o_l2[NET] = stage(G.o_l2[NET], HT * HT * 16 * 64);
    L.o_b2[NET] = stage(G.o_b2[NET], HT * 64);
    L.o_l3[NET] = stage(G.o_l3[NET], ot_net * HT * 16 * 64);
    L.o_b3[NET] = stage(G.o_b3[NET], ot_net * 64);
    L.o_logstd = stage(G.o_logstd, 64);
    L.o_l3t[NET] = stage(G.o_l3t[NET], HT * G.k3s[NET] * 64);
    L.o_l2t[NET] = stage(G.o_l2t[NET], HT * HT * 16 * 64);
    __syncthreads();
    const int lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // wave-uniform: the tile arithmetic stays in scalar registers
    const int col = lane & 31, half = lane >> 5;
    constexpr int S3S = NJ + 1;                            // row stride of the delta3 staging (odd)
    // KX == 1: the tile's observations [64 samples][33], staged once - or (PF) two buffers of pf_rows rows of the tile's inputs:
    // rows [0, obs_dim) observation columns, then NET 0: act_dim action columns, advantage, old log-probability; NET 1: old value, return;
    const int pf_rows = obs_dim + (NET == 0 ? act_dim + 2 : 2) + 2;   // (+ 2: the next tile's row index, low and high words)
    const int XSN = PF ? 2 * pf_rows * PFS : (KX == 1 ? 64 * 33 : 0);
    float *T = lds + lds_used + wave * (32 * 33 + 64 * S3S + XSN);   // transpose scratch, the delta3 staging [64][S3S], observations
    float *S3 = T + 32 * 33;
    float *XS = S3 + 64 * S3S;
    const float onehot = half ? 0.0f : 1.0f;
    const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const long B = a.B, n_tiles = (B + 63) / 64;
    const int mrow = 0;                                     // (the staged layer-1 block holds this net's two row tiles)
    const int k3s = L.k3s[NET];

    // gradient accumulators, alive across the wave's tiles
    f32x16 G1[HT][KX], G2[HT][HT], G3[OT][HT];
    float db2[HT];                                          // bias 2: this lane's unit (col) of tile o, summed over its half's samples
    float db3[NJ], gls[NJ];
    float acc3[HT][16];                                     // value net: per-lane sums of dW3 (see the forward pass)
#pragma unroll
    for (int m = 0; m < HT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc3[m][r] = 0.0f;
    float loss_sum = 0.0f;
#pragma unroll
    for (int o = 0; o < HT; ++o) {
        db2[o] = 0.0f;
#pragma unroll
        for (int k = 0; k < KX; ++k) G1[o][k] = zero;
#pragma unroll
        for (int m = 0; m < HT; ++m) G2[o][m] = zero;
    }
#pragma unroll
    for (int q = 0; q < OT; ++q)
#pragma unroll
        for (int m = 0; m < HT; ++m) G3[q][m] = zero;
#pragma unroll
    for (int j = 0; j < NJ; ++j) { db3[j] = 0.0f; gls[j] = 0.0f; }

    const long tile0 = long(blockIdx.x) * nw + wave, tstep = long(gridDim.x) * nw;
    // (PF) this lane's sample of tile t: position in the minibatch and row in the rollout tensors
    auto pos_of = [&](long t) { const long p = t * 64 + lane; return p < B ? p : B - 1; };
    auto row_of = [&](long t) { const long p = pos_of(t); return a.index ? long(a.index[p]) : p; };
    auto prefetch = [&](long t, long row, int buf) {        // tile t's rows (this lane's sample at `row`) + the sample order of the tile after it
        float *dst = XS + buf * pf_rows * PFS;
        const float *xr = a.obs + row * obs_dim;
        for (int k = 0; k < obs_dim; ++k) dma_dword(xr + k, dst + k * PFS);
        dst += obs_dim * PFS;
        if (NET == 0) {
            const float *ar = a.act + row * act_dim;
            for (int j = 0; j < act_dim; ++j) dma_dword(ar + j, dst + j * PFS);
            dma_dword(a.adv + (a.adv_stats ? row : pos_of(t)), dst + act_dim * PFS);
            dma_dword(a.logp_old + row, dst + (act_dim + 1) * PFS);
            dst += (act_dim + 2) * PFS;
        } else {
            dma_dword(a.val_old + row, dst);
            dma_dword(a.ret + row, dst + PFS);
            dst += 2 * PFS;
        }
        // the next tile's row index travels the same way (two rows: low and high words): an ordinary load in this loop
        // would have the compiler wait - vmcnt(0) - behind the DMAs wherever it moves the loaded register
        if (a.index && t + tstep < n_tiles) {
            const float *ip = reinterpret_cast<const float *>(a.index + pos_of(t + tstep));
            dma_dword(ip, dst);
            dma_dword(ip + 1, dst + PFS);
        }
    };
    // the minibatch's advantage statistics, read once (uniform: scalar registers)
    float adv_mean = 0.0f, adv_istd = 1.0f;
    if (NET == 0 && a.adv_stats) {
        adv_mean = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(a.adv_stats[0])));
        adv_istd = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(a.adv_stats[1])));
    }
    int buf = 0;
    if (PF && tile0 < n_tiles) prefetch(tile0, row_of(tile0), 0);
    const int lane_id = lane;
    for (long tile = tile0; tile < n_tiles; tile += tstep, buf ^= 1) {
        // the lane-derived indices are recomputed per tile from an opaque copy of the lane id: hoisted out of the loop they
        // are a dozen registers the allocator spills, and a scratch reload waits for every DMA issued before it
        int lane = lane_id;
        asm volatile("" : "+v"(lane));
        const int col = lane & 31, half = lane >> 5;
        const float onehot = half ? 0.0f : 1.0f;
        const float *PB = XS + buf * pf_rows * PFS;         // (PF) this tile's inputs
        if (PF) {
            wait_dma();                                     // this tile's rows have landed (issued one tile ago)
            if (tile + tstep < n_tiles) {                   // the next tile's go out now and have this tile's arithmetic to arrive
                long row_n = pos_of(tile + tstep);
                if (a.index) {
                    const unsigned lo = __float_as_uint(PB[(pf_rows - 2) * PFS + lane]), hi = __float_as_uint(PB[(pf_rows - 1) * PFS + lane]);
                    row_n = long((static_cast<unsigned long long>(hi) << 32) | lo);
                }
                prefetch(tile + tstep, row_n, buf ^ 1);
            }
        }
        // ================= forward =================
        long s0 = tile * 64 + col, s1 = s0 + 32;
        s0 = s0 < B ? s0 : B - 1; s1 = s1 < B ? s1 : B - 1;
        if (!PF && a.index) { s0 = a.index[s0]; s1 = a.index[s1]; }
        const float *x0 = a.obs + s0 * obs_dim, *x1 = a.obs + s1 * obs_dim;
        if (KX == 1 && !PF) {
            // the tile's observation rows (gathered when indexed) go to LDS once: lane = sample reads its whole row with
            // all loads in flight together; the K loop of layer 1 and the dW1 operands then come from LDS instead of
            // one dependent global load per step (the indexed minibatch cost 10.4 ms against 7.6 contiguous before)
            long sr = tile * 64 + lane;
            sr = sr < B ? sr : B - 1;
            if (a.index) sr = a.index[sr];
            const float *xr = a.obs + sr * obs_dim;
#pragma unroll
            for (int k0 = 0; k0 < 32; k0 += 8) {                 // eight loads in flight at a time (registers are scarce here)
                float v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = k0 + k < obs_dim ? xr[k0 + k] : (k0 + k == obs_dim ? 1.0f : 0.0f);
#pragma unroll
                for (int k = 0; k < 8; ++k) XS[lane * 33 + k0 + k] = v[k];
            }
            wave_fence();
        }
        f32x16 h1[HT][2], h2[HT][2];
#pragma unroll
        for (int m = 0; m < HT; ++m) { h1[m][0] = zero; h1[m][1] = zero; }
        const float *w1 = lds + L.o_l1 + lane;
        for (int s = 0; s < L.k1s; ++s) {
            const int k = 2 * s + half;
            float b0, b1;
            if (PF) {
                const int kk = k < obs_dim ? k : obs_dim - 1;
                const float v0 = pinned(PB[kk * PFS + col]), v1 = pinned(PB[kk * PFS + 32 + col]);
                const float pad = k == obs_dim ? 1.0f : 0.0f;
                b0 = k < obs_dim ? v0 : pad; b1 = k < obs_dim ? v1 : pad;
            } else if (KX == 1) { b0 = XS[col * 33 + k]; b1 = XS[(32 + col) * 33 + k]; }
            else {
                b0 = k < obs_dim ? x0[k] : (k == obs_dim ? 1.0f : 0.0f);
                b1 = k < obs_dim ? x1[k] : (k == obs_dim ? 1.0f : 0.0f);
            }
#pragma unroll
            for (int m = 0; m < HT; ++m) {
                const float w = w1[((mrow + m) * L.k1s + s) * 64];
                h1[m][0] = mfma(w, b0, h1[m][0]);
                h1[m][1] = mfma(w, b1, h1[m][1]);
            }
        }
#pragma unroll
        for (int m = 0; m < HT; ++m) { tanh_tile(h1[m][0]); tanh_tile(h1[m][1]); }
#pragma unroll
        for (int o = 0; o < HT; ++o) {
            h2[o][0] = zero; h2[o][1] = zero;
            const float *w = lds + L.o_l2[NET] + o * (HT * 16 * 64) + lane;
            mfma_stream<HT * 16>(w, [&](int k) { return h1[k >> 4][0][k & 15]; }, [&](int k) { return h1[k >> 4][1][k & 15]; },
                                 h2[o][0], h2[o][1]);
            const float b = lds[L.o_b2[NET] + o * 64 + lane];
            h2[o][0] = mfma(b, onehot, h2[o][0]);
            h2[o][1] = mfma(b, onehot, h2[o][1]);
            tanh_tile(h2[o][0]); tanh_tile(h2[o][1]);
        }
        float out[NJ];                                      // this lane's sample: output row j
        // The value net has ONE output: as 32-row MFMA tiles its output layer, W3^T delta3 and dW3 would be 31/32
        // padding (134 of the net's 602 MFMAs per tile).  They run on the VALU instead: every lane holds 32 of the 64
        // h2 units of its column's sample, so the output is a 32-term dot product per half-wave plus the other half's.
        float wv[HT][16];                                   // W3[0][unit of (m, r, this half)]
        float bt0 = 0.0f, bt1 = 0.0f;                       // delta3 of this lane's column sample in tile 0 / 1
        if (NET == 1) {
            float p0 = 0.0f, p1 = 0.0f;
#pragma unroll
            for (int m = 0; m < HT; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    wv[m][r] = lds[L.o_l3[NET] + (m * 16 + r) * 64 + 32 * half];      // row 0 of the A operand: lane 0 / 32
                    p0 += wv[m][r] * h2[m][0][r];
                    p1 += wv[m][r] * h2[m][1][r];
                }
            half_swap(p0, p1);                               // (tile 0 | tile 1) x (lower | upper units) -> one sample per lane
            out[0] = p0 + p1 + lds[L.o_b3[NET]];
        }
#pragma unroll
        for (int q = 0; q < (NET == 1 ? 0 : OT); ++q) {
            f32x16 y0 = zero, y1 = zero;
            const float *w = lds + L.o_l3[NET] + q * (HT * 16 * 64) + lane;
            mfma_stream<HT * 16>(w, [&](int k) { return h2[k >> 4][0][k & 15]; }, [&](int k) { return h2[k >> 4][1][k & 15]; }, y0, y1);
            const float b = lds[L.o_b3[NET] + q * 64 + lane];
            y0 = mfma(b, onehot, y0);
            y1 = mfma(b, onehot, y1);
#pragma unroll
            for (int r = 0; r < 16; ++r) {                  // one sample per lane: rows 32 q + U(r) and + 4
                float lo = y0[r], hi = y1[r];
                half_swap(lo, hi);
                if (32 * q + unit_of(r) < NJ) out[32 * q + unit_of(r)] = lo;
                if (32 * q + unit_of(r) + 4 < NJ) out[32 * q + unit_of(r) + 4] = hi;
            }
        }
        // ================= loss derivative of this lane's sample =================
        const long i = tile * 64 + lane;
        const bool live = i < B;
        const long im = live ? i : B - 1;                    // position in the minibatch
        const long ii = PF ? im : (a.index ? long(a.index[im]) : im);     // row in the rollout tensors (PF: the inputs are in PB)
        float d3[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) d3[j] = 0.0f;
        if (NET == 0) {
            float lp = -0.91893853320467274f * float(act_dim);
            float z[NJ], iv[NJ];
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                z[j] = 0.0f; iv[j] = 0.0f;
                if (j < act_dim) {
                    const float ls = lds[L.o_logstd + j];
                    iv[j] = __expf(-2.0f * ls);
                    z[j] = (PF ? PB[(obs_dim + j) * PFS + lane] : a.act[ii * act_dim + j]) - out[j];
                    lp -= 0.5f * z[j] * z[j] * iv[j] + ls;
                }
            }
            const float araw = PF ? PB[(obs_dim + act_dim) * PFS + lane] : (a.adv_stats ? a.adv[ii] : a.adv[im]);
            const float A = (araw - adv_mean) * adv_istd;
            const float ratio = __expf(lp - (PF ? PB[(obs_dim + act_dim + 1) * PFS + lane] : a.logp_old[ii]));
            const float rc = __builtin_amdgcn_fmed3f(ratio, 1.0f - a.cliprange, 1.0f + a.cliprange);
            const float t1 = -A * ratio, t2 = -A * rc;
            const float g = live ? (t1 >= t2 ? -A : 0.0f) * ratio * a.inv_B : 0.0f;      // dL / dlogp
            if (live) loss_sum += fmaxf(t1, t2) * a.inv_B;
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                if (j < act_dim) {
                    d3[j] = g * z[j] * iv[j];
                    gls[j] += g * (z[j] * z[j] * iv[j] - 1.0f);
                }
        } else {
            const float v = out[0], vo = PF ? PB[obs_dim * PFS + lane] : a.val_old[ii], R = PF ? PB[(obs_dim + 1) * PFS + lane] : a.ret[ii];
            const float dv = v - vo, vc = vo + __builtin_amdgcn_fmed3f(dv, -a.cliprange, a.cliprange);
            const float e1 = (v - R) * (v - R), e2 = (vc - R) * (vc - R);
            const float dvl = e1 >= e2 ? (v - R) : (fabsf(dv) < a.cliprange ? (vc - R) : 0.0f);
            d3[0] = live ? a.vf_coef * a.inv_B * dvl : 0.0f;
            if (live) loss_sum += 0.5f * fmaxf(e1, e2) * a.inv_B;
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) db3[j] += d3[j];
        // delta3 staged [sample][row] for the transposed (row-on-lane) reads of dW3
        if (NET == 0) {
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                if (j < n_out) S3[lane * S3S + j] = d3[j];
        } else {
            bt0 = d3[0]; bt1 = d3[0];
            half_swap(bt0, bt1);                             // this lane's column sample of tile 0 / tile 1
        }
        // ================= delta2 = (W3^T delta3) (1 - h2^2) =================
        f32x16 d2[HT][2];
#pragma unroll
        for (int m = 0; m < HT; ++m) { d2[m][0] = zero; d2[m][1] = zero; }
        if (NET == 1) {
#pragma unroll
            for (int m = 0; m < HT; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    d2[m][0][r] = wv[m][r] * bt0;
                    d2[m][1][r] = wv[m][r] * bt1;
                    acc3[m][r] += bt0 * h2[m][0][r] + bt1 * h2[m][1][r];     // dW3, reduced over the lanes at the end
                }
        }
#pragma unroll
        for (int s = 0; s < (NET == 1 ? 0 : NJ / 2); ++s)
            if (s < k3s) {                                   // K pair = outputs (2 s, 2 s + 1)
                float b0 = d3[2 * s], b1 = d3[2 * s + 1];
                half_swap(b0, b1);                           // b0: column tile 0, b1: column tile 1
#pragma unroll
                for (int m = 0; m < HT; ++m) {
                    const float w = lds[L.o_l3t[NET] + (m * k3s + s) * 64 + lane];
                    d2[m][0] = mfma(w, b0, d2[m][0]);
                    d2[m][1] = mfma(w, b1, d2[m][1]);
                }
            }
#pragma unroll
        for (int m = 0; m < HT; ++m)
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) d2[m][t][r] *= 1.0f - h2[m][t][r] * h2[m][t][r];
        wave_fence();                                        // S3 written above is read below
        // ================= dW3 += delta3 h2^T, per column tile =================
#pragma unroll
        for (int t = 0; t < (NET == 1 ? 0 : 2); ++t) {
            f32x16 h2T[HT];
#pragma unroll
            for (int m = 0; m < HT; ++m) h2T[m] = transpose_tile(h2[m][t], T, col, half);
#pragma unroll
            for (int q = 0; q < OT; ++q) {
                const int j = 32 * q + col;                  // row on this lane
                const int jj = j < n_out ? j : 0;            // (an unconditional read + select: a guarded read is a branch per element,
                                                             //  32 basic blocks that the MFMAs cannot be scheduled across)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float d3raw = pinned(S3[(32 * t + unit_of(r) + 4 * half) * S3S + jj]);
                    const float d3t = j < n_out ? d3raw : 0.0f;
#pragma unroll
                    for (int m = 0; m < HT; ++m) G3[q][m] = mfma(d3t, h2T[m][r], G3[q][m]);
                }
            }
        }
        // ================= delta1 = (W2^T delta2) (1 - h1^2) =================
        f32x16 d1[HT][2];
#pragma unroll
        for (int ip = 0; ip < HT; ++ip) {
            d1[ip][0] = zero; d1[ip][1] = zero;
            const float *w = lds + L.o_l2t[NET] + ip * (HT * 16 * 64) + lane;
            mfma_stream<HT * 16>(w, [&](int k) { return d2[k >> 4][0][k & 15]; }, [&](int k) { return d2[k >> 4][1][k & 15]; },
                                 d1[ip][0], d1[ip][1]);
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) d1[ip][t][r] *= 1.0f - h1[ip][t][r] * h1[ip][t][r];
        }
        // ================= dW2 += delta2 h1^T, db2, dW1 += delta1 [obs | 1]^T =================
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            f32x16 h1T[HT], dT[HT];
#pragma unroll
            for (int m = 0; m < HT; ++m) { h1T[m] = transpose_tile(h1[m][t], T, col, half); dT[m] = transpose_tile(d2[m][t], T, col, half); }
            // bias 2 from the transposed delta2 (units on the lanes, samples in the registers): one accumulator per row tile
            // instead of a 16-register tile of per-sample sums
#pragma unroll
            for (int m = 0; m < HT; ++m) {
                float sum = 0.0f;
#pragma unroll
                for (int r = 0; r < 16; ++r) sum += dT[m][r];
                db2[m] += sum;
            }
#pragma unroll
            for (int o = 0; o < HT; ++o)
#pragma unroll
                for (int m = 0; m < HT; ++m)
#pragma unroll
                    for (int r = 0; r < 16; ++r) G2[o][m] = mfma(dT[o][r], h1T[m][r], G2[o][m]);
#pragma unroll
            for (int m = 0; m < HT; ++m) dT[m] = transpose_tile(d1[m][t], T, col, half);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                long sn = tile * 64 + 32 * t + unit_of(r) + 4 * half;          // sample of this K slot
                sn = sn < B ? sn : B - 1;
                if (KX != 1 && a.index) sn = a.index[sn];
#pragma unroll
                for (int kx = 0; kx < KX; ++kx) {
                    const int k = 32 * kx + col;
                    float xv;
                    if (PF) {
                        const float v = pinned(PB[(col < obs_dim ? col : obs_dim - 1) * PFS + 32 * t + unit_of(r) + 4 * half]);
                        xv = col < obs_dim ? v : (col == obs_dim ? 1.0f : 0.0f);
                    } else
                        xv = KX == 1 ? XS[(32 * t + unit_of(r) + 4 * half) * 33 + col]
                                     : (k < obs_dim ? a.obs[sn * obs_dim + k] : (k == obs_dim ? 1.0f : 0.0f));
#pragma unroll
                    for (int o = 0; o < HT; ++o) G1[o][kx] = mfma(dT[o][r], xv, G1[o][kx]);
                }
            }
        }
    }
    // ================= this wave's partial gradient, torch parameter order =================
    const GOff g = goff_of(obs_dim, n_out);
    float *P = a.partials + (long(blockIdx.x) * nw + wave) * a.gstride;
    for (int k = lane; k < a.gstride; k += 64) P[k] = 0.0f;
    wave_fence();
    __builtin_amdgcn_s_waitcnt(0);
#pragma unroll
    for (int o = 0; o < HT; ++o)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = 32 * o + unit_of(r) + 4 * half;              // out unit
#pragma unroll
            for (int m = 0; m < HT; ++m) P[g.w2 + row * H + 32 * m + col] = G2[o][m][r];
#pragma unroll
            for (int kx = 0; kx < KX; ++kx) {
                const int k = 32 * kx + col;
                if (k < obs_dim) P[g.w1 + row * obs_dim + k] = G1[o][kx][r];
                else if (k == obs_dim) P[g.b1 + row] = G1[o][kx][r];
            }
        }
#pragma unroll
    for (int o = 0; o < HT; ++o) {                                       // bias 2: unit 32 o + col, the two halves' samples
        const float v = db2[o] + __shfl_xor(db2[o], 32, 64);
        if (half == 0) P[g.b2 + 32 * o + col] = v;
    }
    if (NET == 1) {
#pragma unroll
        for (int m = 0; m < HT; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = acc3[m][r];                                    // sum over the 32 samples-lanes of the half-wave
#pragma unroll
                for (int off = 16; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
                if (col == 0) P[g.w3 + 32 * m + unit_of(r) + 4 * half] = v;
            }
    }
#pragma unroll
    for (int q = 0; q < (NET == 1 ? 0 : OT); ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = 32 * q + unit_of(r) + 4 * half;              // output
            if (row < n_out) {
#pragma unroll
                for (int m = 0; m < HT; ++m) P[g.w3 + row * H + 32 * m + col] = G3[q][m][r];
            }
        }
#pragma unroll
    for (int j = 0; j < NJ; ++j)
        if (j < n_out) {
            float v = db3[j], w = gls[j];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) { v += __shfl_xor(v, off, 64); w += __shfl_xor(w, off, 64); }
            if (lane == 0) { P[g.b3 + j] = v; P[g.ls + j] = NET == 0 ? w : 0.0f; }
        }
    float ls = loss_sum;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) ls += __shfl_xor(ls, off, 64);
    if (lane == 0) P[g.loss] = ls;
    // the workgroup's waves fold their partials into wave 0's (fixed order: bit-reproducible), so the reduction
    // kernel reads one partial per workgroup instead of one per wave
    __syncthreads();
    float *P0 = a.partials + long(blockIdx.x) * nw * a.gstride;
    for (int k = threadIdx.x; k < a.gstride; k += blockDim.x) {
        float v = P0[k];
        for (int w = 1; w < nw; ++w) v += P0[w * a.gstride + k];
        P0[k] = v;
    }
}

// out[k] = sum over the partials (one per workgroup, `gstride` apart)
__global__ void reduce_partials_kernel(const float *__restrict__ partials, int n_waves, int gstride, int n, float *__restrict__ out) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
    int w = 0;
    for (; w + 3 < n_waves; w += 4) {
        s0 += partials[long(w) * gstride + k]; s1 += partials[long(w + 1) * gstride + k];
        s2 += partials[long(w + 2) * gstride + k]; s3 += partials[long(w + 3) * gstride + k];
    }
    for (; w < n_waves; ++w) s0 += partials[long(w) * gstride + k];
    out[k] = (s0 + s1) + (s2 + s3);
}

// generalised advantage estimation over a [T][N] rollout, one env per thread, backwards in time (ppo.py: gae())
__global__ void gae_kernel(const float *__restrict__ rew, const float *__restrict__ val, const float *__restrict__ done,
                           const float *__restrict__ last_val, float gamma, float lam, float *__restrict__ adv,
                           float *__restrict__ ret, int T, long n) {
    const long i = long(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float next_value = last_val[i], last = 0.0f;
    for (int t = T - 1; t >= 0; --t) {
        const float nonterminal = 1.0f - done[t * n + i], v = val[t * n + i];
        const float delta = rew[t * n + i] + gamma * next_value * nonterminal - v;
        last = delta + gamma * lam * nonterminal * last;
        adv[t * n + i] = last;
        ret[t * n + i] = last + v;
        next_value = v;
    }
}

constexpr int WAVES_PER_BLOCK = 4;
// (every MI355X of a node has the same CU count, so the workspace size a caller asks for before it names a device -
// rp_ppo_workspace_floats - and the grid of the launch agree; the count is the current device's)
long grad_blocks(long B) {
    const long tiles = (B + 63) / 64;
    long blocks = (tiles + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK;
    int dev = 0;
    (void)hipGetDevice(&dev);
    const long cap = cu_count(dev);
    return blocks < cap ? blocks : cap;
}

// dynamic LDS of one gradient-kernel instance: the staged operand blocks of one net + per-wave scratch
// (pf_rows > 0: the prefetching form's two input buffers per wave instead of the observation staging)
size_t grad_lds_bytes(const Layout &L, int net, int nj, int kx_inst, int pf_rows = 0) {
    const int ot = net == 0 ? L.ot_pi : 1;
    const size_t w = size_t(HT) * L.k1s * 64 + 2 * size_t(HT * HT * 16 * 64) + HT * 64 + size_t(ot) * (HT * 16 * 64 + 64) + 64 +
                     size_t(HT) * L.k3s[net] * 64;
    const size_t xs = pf_rows > 0 ? size_t(2) * pf_rows * PFS : (kx_inst == 1 ? 64 * 33 : 0);
    return sizeof(float) * (w + WAVES_PER_BLOCK * (32 * 33 + 64 * (nj + 1) + xs));
}
// the instance pair rp_ppo_grad_dev picks: the reference's robot class (obs <= 31, up to 8 actions) or the general
// one (obs <= 63, up to 64 actions); false if the general instance of the action net does not fit the 160 KB of LDS
bool grad_fits(int obs_dim, int act_dim) {
    const Layout L = layout_of(obs_dim, act_dim);
    const bool small = obs_dim + 1 <= 32 && act_dim <= 8;
    return grad_lds_bytes(L, 0, small ? 8 : 64, small ? 1 : 2) <= 160 * 1024;
}

template <int NET, int KX, int NJ, bool PF = false>
int launch_grad(const TrainArgs &a, long blocks, size_t lds, int dev, hipStream_t stream) {
    // the opt-in above 64 KB of dynamic LDS is per kernel AND per device (mlp_common.hpp: grant_lds)
    constexpr int kernel_id = 1 + NET * 2 + (KX - 1) + (PF ? 4 : 0);
    if (int rc = grant_lds(reinterpret_cast<const void *>(&mlp_grad_kernel<NET, KX, NJ, PF>), kernel_id, dev, lds)) return rc;
    hipLaunchKernelGGL((mlp_grad_kernel<NET, KX, NJ, PF>), dim3(unsigned(blocks)), dim3(64 * WAVES_PER_BLOCK), lds, stream, a);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(RP_EHIP, std::string("mlp_grad_kernel: ") + hipGetErrorString(e));
    return RP_OK;
}

}  // namespace

extern "C" {

int64_t rp_train_packed_floats(int obs_dim, int act_dim) {
    if (rp_packed_floats(obs_dim, act_dim) < 0) return RP_EUNSUPPORTED;
    if (obs_dim + 1 > 64) return fail(RP_EUNSUPPORTED, "the gradient kernel supports obs_dim <= 63");
    if (!grad_fits(obs_dim, act_dim))
        return fail(RP_EUNSUPPORTED, "policy too large for the LDS-resident gradient kernel (operands + scratch > 160 KB)");
    return layout_of(obs_dim, act_dim).total_train;
}

int rp_pack_train(const rp_mlp_params *p, int obs_dim, int act_dim, float *out) {
    if (rp_train_packed_floats(obs_dim, act_dim) < 0) return RP_EUNSUPPORTED;
    const int rc = rp_pack(p, obs_dim, act_dim, out);
    if (rc) return rc;
    const Layout L = layout_of(obs_dim, act_dim);
    std::memset(out + L.total, 0, sizeof(float) * size_t(L.total_train - L.total));
    const float *w2[2] = {p->pi_w2, p->vf_w2}, *w3[2] = {p->pi_w3, p->vf_w3};
    for (int net = 0; net < 2; ++net) {
        const int n_out = net == 0 ? act_dim : 1;
        for (int m = 0; m < HT; ++m)
            for (int s = 0; s < L.k3s[net]; ++s)
                for (int l = 0; l < 64; ++l) {
                    const int unit = 32 * m + (l & 31), j = 2 * s + (l >> 5);
                    out[L.o_l3t[net] + (m * L.k3s[net] + s) * 64 + l] = j < n_out ? w3[net][j * H + unit] : 0.0f;
                }
        for (int ip = 0; ip < HT; ++ip)
            for (int o = 0; o < HT; ++o)
                for (int r = 0; r < 16; ++r)
                    for (int l = 0; l < 64; ++l) {
                        const int in = 32 * ip + (l & 31), outu = 32 * o + unit_of(r) + 4 * (l >> 5);
                        out[L.o_l2t[net] + ((ip * HT + o) * 16 + r) * 64 + l] = w2[net][outu * H + in];
                    }
    }
    return RP_OK;
}

int rp_gae_dev(const float *d_rew, const float *d_val, const float *d_done, const float *d_last_val, float gamma, float lam,
               float *d_adv, float *d_ret, int n_steps, int64_t n_envs, void *stream) {
    if (!d_rew || !d_val || !d_done || !d_last_val || !d_adv || !d_ret) return fail(RP_EINVAL, "null argument");
    if (n_steps < 1 || n_envs < 1) return fail(RP_EINVAL, "n_steps and n_envs must be >= 1");
    DeviceScope scope(d_rew); if (scope.rc) return scope.rc;
    hipLaunchKernelGGL(gae_kernel, dim3(unsigned((n_envs + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), d_rew,
                       d_val, d_done, d_last_val, gamma, lam, d_adv, d_ret, n_steps, long(n_envs));
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(RP_EHIP, std::string("gae_kernel: ") + hipGetErrorString(e));
    return RP_OK;
}

// instances: the reference's robot class (obs <= 31, up to 8 actions; its inputs prefetched by LDS-DMA when the two
// buffers per wave fit beside the operands - ROBOY_POLICY_PREFETCH=0 keeps the loads at the start of each tile) and the
// general one (obs <= 63, 64 actions)
int rp_grad_form(int obs_dim, int act_dim) {
    if (rp_train_packed_floats(obs_dim, act_dim) < 0) return RP_EUNSUPPORTED;
    const Layout L = layout_of(obs_dim, act_dim);
    if (!((obs_dim + 1 + 31) / 32 == 1 && act_dim <= 8)) return 0;
    static const bool prefetch_off = [] { const char *e = getenv("ROBOY_POLICY_PREFETCH"); return e && e[0] == '0'; }();
    const bool fits = grad_lds_bytes(L, 0, 8, 1, obs_dim + act_dim + 4) <= 160 * 1024 && grad_lds_bytes(L, 1, 8, 1, obs_dim + 4) <= 160 * 1024;
    return !prefetch_off && fits ? 2 : 1;
}

int64_t rp_grad_floats(int obs_dim, int act_dim) {
    if (rp_train_packed_floats(obs_dim, act_dim) < 0) return RP_EUNSUPPORTED;
    return 2 * int64_t(gstride_of(obs_dim, act_dim));
}

int64_t rp_ppo_workspace_floats(int obs_dim, int act_dim, int64_t batch) {
    if (rp_train_packed_floats(obs_dim, act_dim) < 0 || batch < 1) return RP_EUNSUPPORTED;
    return 2 * grad_blocks(batch) * WAVES_PER_BLOCK * int64_t(gstride_of(obs_dim, act_dim));
}

int rp_ppo_grad_dev(const float *d_packed_train, const float *d_obs, const float *d_act, const float *d_adv,
                    const float *d_adv_stats, const float *d_logp_old, const float *d_val_old, const float *d_ret,
                    const int64_t *d_index, int64_t batch, int obs_dim, int act_dim, float cliprange, float vf_coef,
                    float *d_grad, float *d_workspace, void *stream) {
    if (!d_packed_train || !d_obs || !d_act || !d_adv || !d_logp_old || !d_val_old || !d_ret || !d_grad || !d_workspace)
        return fail(RP_EINVAL, "null argument");
    if (batch < 1) return fail(RP_EINVAL, "batch must be >= 1");
    if (rp_train_packed_floats(obs_dim, act_dim) < 0) return RP_EUNSUPPORTED;
    int dev = 0;
    DeviceScope scope(d_packed_train); if (scope.rc) return scope.rc; dev = scope.dev;      // the blob's device is the device of the call
    const Layout L = layout_of(obs_dim, act_dim);
    const int gs = gstride_of(obs_dim, act_dim);
    const long blocks = grad_blocks(batch), waves = blocks * WAVES_PER_BLOCK;
    auto lds_of = [&](int net, int nj, int kx_inst, int pf_rows = 0) { return grad_lds_bytes(L, net, nj, kx_inst, pf_rows); };
    hipStream_t st = static_cast<hipStream_t>(stream);
    TrainArgs a;
    a.packed = d_packed_train; a.obs = d_obs; a.act = d_act; a.adv = d_adv; a.adv_stats = d_adv_stats; a.logp_old = d_logp_old; a.val_old = d_val_old;
    a.ret = d_ret; a.index = reinterpret_cast<const long long *>(d_index); a.B = batch; a.obs_dim = obs_dim; a.act_dim = act_dim; a.gstride = gs; a.cliprange = cliprange;
    a.vf_coef = vf_coef; a.inv_B = 1.0f / float(batch);
    const int kx = (obs_dim + 1 + 31) / 32;
    int rc;
    a.partials = d_workspace;
    const int form = rp_grad_form(obs_dim, act_dim);                     // 2: small + prefetch, 1: small, 0: general
    const bool small = form >= 1, pf = form == 2;
    const int pf0 = obs_dim + act_dim + 4, pf1 = obs_dim + 4;           // rows of one input buffer (mlp_grad_kernel: pf_rows)
    if (pf) rc = launch_grad<0, 1, 8, true>(a, blocks, lds_of(0, 8, 1, pf0), dev, st);
    else if (small) rc = launch_grad<0, 1, 8>(a, blocks, lds_of(0, 8, 1), dev, st);
    else rc = launch_grad<0, 2, 64>(a, blocks, lds_of(0, 64, 2), dev, st);
    if (rc) return rc;
    a.partials = d_workspace + waves * gs;
    if (pf) rc = launch_grad<1, 1, 8, true>(a, blocks, lds_of(1, 8, 1, pf1), dev, st);
    else if (small) rc = launch_grad<1, 1, 8>(a, blocks, lds_of(1, 8, 1), dev, st);
    else rc = launch_grad<1, 2, 8>(a, blocks, lds_of(1, 8, 2), dev, st);
    if (rc) return rc;
    for (int net = 0; net < 2; ++net) {
        hipLaunchKernelGGL(reduce_partials_kernel, dim3((gs + 255) / 256), dim3(256), 0, st, d_workspace + net * waves * gs,
                           int(blocks), WAVES_PER_BLOCK * gs, gs, d_grad + net * gs);       // one folded partial per workgroup
    }
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(RP_EHIP, std::string("reduce_partials_kernel: ") + hipGetErrorString(e));
    return RP_OK;
}

}  // extern "C"
