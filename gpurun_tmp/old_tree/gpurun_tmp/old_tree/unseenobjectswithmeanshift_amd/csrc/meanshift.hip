// Classic vMF mean-shift clustering over unit embeddings (see include/msm_hip.h: msm_ms_*).
//
// Reference: lib/utils/mean_shift.py -- select_smart_seeds :128-189 (farthest-point seeding with
// d = 0.5*(1 - x.s), first-max argmax), seed_hill_climbing_ball :79-109 (W = exp(kappa Z X^T),
// Z <- normalize(W X), no max subtraction), the assignment / relabel tail of mean_shift_smart_init
// :206-229.  connected_components (:41-76) is sequential over <= a few hundred seeds and stays on
// the host (unseenobjectswithmeanshift_amd/mean_shift.py).
//
// What bounds what (n = 307200, d = 64, S = 100):
//   seeding   : S passes over X (78.6 MB each) -> HBM-bound streaming + a two-stage argmax; the
//               reference also keeps an (n, S) distance matrix and re-reduces it every pass, here
//               a running minimum (4 B/point) carries the same values exactly (min is exact);
//   hill climb: 4*S*n*d FLOP per iteration on v_mfma_f32_16x16x4_f32 with X streamed once per
//               iteration; the (S, n) kernel matrix W (123 MB) is never materialised: a wave
//               turns a 16-point block into exp() weights in registers, already in A-operand
//               layout for the W X product;
//   assignment: one more X pass, S^T tiles + an in-register first-min argmin.
#include "bf16.h"
#include "common.h"
#include <cstdlib>

namespace msm {

constexpr int MS_D = 64;
constexpr int MS_SB = 19;              // up to 19 seed blocks of 16 -> S <= 304
constexpr int MS_CH = 8;               // seed blocks handled per kernel instance (128 seeds)
constexpr int SZ = MS_D + 4;           // LDS row stride in floats: 16-byte aligned rows, 17 slots apart -> b128 reads spread over the banks

// ------------------------------------------------------------------------------------------------
// seeding
// ------------------------------------------------------------------------------------------------
// One farthest-point step = ONE launch, no inter-workgroup hand-off:
//   nearest[i] = min(nearest[i], 0.5*(1 - X[i].X[winner(step-1)])), and the step's winner (first argmax) is
//   folded into a single 64-bit atomicMax key  (order-preserving bits of the value) << 32 | (~index),
//   so the larger value wins and, on ties, the smaller index -- exactly torch.argmax.  The NEXT launch
//   decodes keys[step-1]; the kernel boundary is the only synchronisation.
// Mapping: 16 lanes own 16 consecutive rows.  Each lane loads its 16-byte column chunk of all 16 rows
// (16 independent loads in flight), and a 4-stage halving butterfly (15 shuffles) leaves lane j with the dot
// product of row j, so the nearest[] update and the running argmax use every lane and 64-byte accesses.
__device__ __forceinline__ unsigned int ordered_bits(float v) {
    const unsigned int u = __float_as_uint(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

template <bool SMALL>
__global__ __launch_bounds__(256) void ms_seed_step_kernel(const float* __restrict__ X, int n,
                                                           unsigned long long* __restrict__ keys, int step,
                                                           float* __restrict__ nearest) {
    __shared__ float4 seed4[16];
    __shared__ unsigned long long red[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15;         // column chunk while loading, row within the group after the butterfly
    const int grp = lane >> 4;
    const int rows_per_block = (((n + gridDim.x - 1) / gridDim.x) + 15) & ~15;
    const int r0 = blockIdx.x * rows_per_block, r1 = min(n, r0 + rows_per_block);
    int base = r0 + (wave * 4 + grp) * 16;
    // The last group of the array is slid back to rows [n-16, n): re-processing a row is idempotent (same min,
    // same key), so no per-row clamps and the 16 row loads share one address register + immediate offsets.
    // (n < 16 keeps the rows clamped instead: SMALL.)
    // program order = issue order: previous winner's key, this pass's 16 rows + nearest[], then the winner's row,
    // so the two dependent latencies (key -> seed row) overlap the streaming loads
    const unsigned long long prev = keys[step - 1];
    float4 x[16];
    int gb = SMALL ? base : min(base, n - 16);
    {
        const float* src = X + (int64_t)gb * MS_D + j * 4;
#pragma unroll
        for (int i = 0; i < 16; ++i)
            x[i] = *reinterpret_cast<const float4*>(SMALL ? X + (int64_t)min(gb + i, n - 1) * MS_D + j * 4 : src + i * MS_D);
    }
    float near = (step > 1) ? nearest[min(gb + j, n - 1)] : INFINITY;
    const unsigned int cur = 0xFFFFFFFFu - (unsigned int)(prev & 0xFFFFFFFFull);
    if (tid < 16) seed4[tid] = *reinterpret_cast<const float4*>(X + (int64_t)cur * MS_D + tid * 4);
    __syncthreads();
    const float4 s = seed4[j];
    unsigned long long best = 0ull;
    for (int pass = 0; base < r1; base += 256, ++pass) {
        if (pass > 0) {   // multi-pass launches (n > 512 Ki rows); other resident waves cover this latency
            gb = SMALL ? base : min(base, n - 16);
            const float* src = X + (int64_t)gb * MS_D + j * 4;
#pragma unroll
            for (int i = 0; i < 16; ++i)
                x[i] = *reinterpret_cast<const float4*>(SMALL ? X + (int64_t)min(gb + i, n - 1) * MS_D + j * 4 : src + i * MS_D);
            if (step > 1) near = nearest[min(gb + j, n - 1)];
        }
        float p[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) p[i] = x[i].x * s.x + x[i].y * s.y + x[i].z * s.z + x[i].w * s.w;
        const float near_now = near;
        const int row = gb + j;
        // halving butterfly: after the stage with mask m, lanes with (j & m) keep the upper half of the rows
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const bool hi = j & 8;
            const float send = hi ? p[i] : p[i + 8];
            const float keep = hi ? p[i + 8] : p[i];
            p[i] = keep + wave_xor_dpp8(send);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool hi = j & 4;
            const float send = hi ? p[i] : p[i + 4];
            const float keep = hi ? p[i + 4] : p[i];
            p[i] = keep + wave_xor_dpp4(send);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const bool hi = j & 2;
            const float send = hi ? p[i] : p[i + 2];
            const float keep = hi ? p[i + 2] : p[i];
            p[i] = keep + wave_xor_dpp2(send);
        }
        {
            const bool hi = j & 1;
            const float send = hi ? p[0] : p[1];
            const float keep = hi ? p[1] : p[0];
            p[0] = keep + wave_xor_dpp1(send);
        }
        if (row < n) {
            const float d = fminf(near_now, 0.5f * (1.0f - p[0]));
            nearest[row] = d;
            const unsigned long long key = ((unsigned long long)ordered_bits(d) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned int)row);
            best = key > best ? key : best;
        }
    }
    best = wave_max_u64(best);
    if (lane == 0) red[wave] = best;
    __syncthreads();
    if (tid == 0) {
        unsigned long long b = red[0];
        b = red[1] > b ? red[1] : b;
        b = red[2] > b ? red[2] : b;
        b = red[3] > b ? red[3] : b;
        if (b) atomicMax(&keys[step], b);
    }
}

// ---- the same step on a bf16 copy of X (precision "bf16": BASELINE configs[4], n = 1 228 800, 300 seeds) ------------------------
// At that size seeding is S passes over X and nothing else: 300 x 314 MB = 94 GB, 15.5 ms at 6.1 TB/s -- the fp32 kernel sits on
// the HBM wall (0.76 of the 8 TB/s peak, 0.97 of what a copy reaches).  The only way through is fewer bytes: a bf16 copy of X
// (made once per clustering, ms_pack_bf16_kernel) is 157 MB, half the stream and small enough for the 256 MB Infinity Cache to
// hold between the steps.  Distances are then those of the ROUNDED points (|d - d_fp32| < 2^-9): the seeds are a farthest-point
// set of the same clusters, not the same indices (labels identical up to permutation on planted clusters) -- tests
// compare this mode by cluster structure, the fp32 / f32_split modes exactly.
// Same mapping as ms_seed_step_kernel: 16 lanes own 16 rows; a lane loads its 8-byte chunk (4 bf16) of each row, the dot product
// is two v_dot2_f32_bf16 per row against the winner's chunk, the butterfly and the (value, ~index) key are unchanged.
typedef __bf16 ms_bf2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float dot4_bf16(uint2 a, uint2 b) {
    float acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(ms_bf2, a.x), __builtin_bit_cast(ms_bf2, b.x), 0.f, false);
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(ms_bf2, a.y), __builtin_bit_cast(ms_bf2, b.y), acc, false);
}

__global__ __launch_bounds__(256) void ms_pack_bf16_kernel(const float* __restrict__ X, int n, int n_pad, uint16_t* __restrict__ Xb) {
    const int64_t total4 = (int64_t)n_pad * (MS_D / 4);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (int64_t)gridDim.x * 256) {
        const float4 v = (i < (int64_t)n * (MS_D / 4)) ? *reinterpret_cast<const float4*>(X + i * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<u32x2b*>(Xb + i * 4) = __builtin_bit_cast(u32x2b, pack4(v.x, v.y, v.z, v.w));
    }
}

// n >= 16 (the caller takes the fp32 kernel below that)
__global__ __launch_bounds__(256) void ms_seed_step_bf16_kernel(const uint16_t* __restrict__ Xb, int n, unsigned long long* __restrict__ keys,
                                                                int step, float* __restrict__ nearest) {
    __shared__ uint2 seed2[16];
    __shared__ unsigned long long red[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, grp = lane >> 4;
    const int rows_per_block = (((n + gridDim.x - 1) / gridDim.x) + 15) & ~15;
    const int r0 = blockIdx.x * rows_per_block, r1 = min(n, r0 + rows_per_block);
    int base = r0 + (wave * 4 + grp) * 16;
    const unsigned long long prev = keys[step - 1];
    uint2 x[16];
    int gb = min(base, n - 16);
    {
        const uint16_t* src = Xb + (int64_t)gb * MS_D + j * 4;
#pragma unroll
        for (int i = 0; i < 16; ++i) x[i] = *reinterpret_cast<const uint2*>(src + i * MS_D);
    }
    float near = (step > 1) ? nearest[min(gb + j, n - 1)] : INFINITY;
    const unsigned int cur = 0xFFFFFFFFu - (unsigned int)(prev & 0xFFFFFFFFull);
    if (tid < 16) seed2[tid] = *reinterpret_cast<const uint2*>(Xb + (int64_t)cur * MS_D + tid * 4);
    __syncthreads();
    const uint2 s = seed2[j];
    unsigned long long best = 0ull;
    for (int pass = 0; base < r1; base += 256, ++pass) {
        if (pass > 0) {
            gb = min(base, n - 16);
            const uint16_t* src = Xb + (int64_t)gb * MS_D + j * 4;
#pragma unroll
            for (int i = 0; i < 16; ++i) x[i] = *reinterpret_cast<const uint2*>(src + i * MS_D);
            if (step > 1) near = nearest[min(gb + j, n - 1)];
        }
        float p[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) p[i] = dot4_bf16(x[i], s);
        const float near_now = near;
        const int row = gb + j;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const bool hi = j & 8;
            const float send = hi ? p[i] : p[i + 8];
            const float keep = hi ? p[i + 8] : p[i];
            p[i] = keep + wave_xor_dpp8(send);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool hi = j & 4;
            const float send = hi ? p[i] : p[i + 4];
            const float keep = hi ? p[i + 4] : p[i];
            p[i] = keep + wave_xor_dpp4(send);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const bool hi = j & 2;
            const float send = hi ? p[i] : p[i + 2];
            const float keep = hi ? p[i + 2] : p[i];
            p[i] = keep + wave_xor_dpp2(send);
        }
        {
            const bool hi = j & 1;
            const float send = hi ? p[0] : p[1];
            const float keep = hi ? p[1] : p[0];
            p[0] = keep + wave_xor_dpp1(send);
        }
        if (row < n) {
            const float d = fminf(near_now, 0.5f * (1.0f - p[0]));
            nearest[row] = d;
            const unsigned long long key = ((unsigned long long)ordered_bits(d) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned int)row);
            best = key > best ? key : best;
        }
    }
    best = wave_max_u64(best);
    if (lane == 0) red[wave] = best;
    __syncthreads();
    if (tid == 0) {
        unsigned long long b = red[0];
        b = red[1] > b ? red[1] : b;
        b = red[2] > b ? red[2] : b;
        b = red[3] > b ? red[3] : b;
        if (b) atomicMax(&keys[step], b);
    }
}

// ---- persistent seeding for maps that fit the register file -----------------------------------------------------
// All S-1 farthest-point steps in ONE launch: X (n x 64 fp32 = 79 MB at 640x480) is read once into VGPRs -- a workgroup
// of 8 waves holds 512*NG rows, 200 workgroups hold the map -- and every step is a dot product against the previous
// winner's row from registers, the same butterfly and (value, ~index) atomicMax key as ms_seed_step_kernel (so the
// selected indices are bit-identical), and an all-to-all exchange of the workgroups' candidates through data-tagged 8-byte
// granules (see the loop).  The per-step cost is that exchange (5.6 us) instead of a 79 MB stream (~22 us).
// Safety: the grid never exceeds the number of CUs (one workgroup per CU is guaranteed by the launch bounds), so all
// workgroups are co-resident; every wait is bounded and raises `status[1]`, which makes every workgroup leave and the
// finish kernel report -1 indices instead of hanging the queue.
constexpr int PS_W = 8;                      // waves per persistent workgroup
constexpr int PS_MAXWG = 256;                // workgroups of the persistent launch (one per CU at most)
constexpr unsigned PS_SPIN_LIMIT = 1u << 20; // polls (with s_sleep) before giving up: well under a second

// lane j of a 16-lane group ends with the full dot product of row j: p[i] holds this lane's partial of row i on entry
__device__ __forceinline__ float butterfly16(float (&p)[16], int j) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const bool hi = j & 8;
        const float send = hi ? p[i] : p[i + 8];
        const float keep = hi ? p[i + 8] : p[i];
        p[i] = keep + wave_xor_dpp8(send);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const bool hi = j & 4;
        const float send = hi ? p[i] : p[i + 4];
        const float keep = hi ? p[i + 4] : p[i];
        p[i] = keep + wave_xor_dpp4(send);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const bool hi = j & 2;
        const float send = hi ? p[i] : p[i + 2];
        const float keep = hi ? p[i + 2] : p[i];
        p[i] = keep + wave_xor_dpp2(send);
    }
    const bool hi = j & 1;
    const float send = hi ? p[0] : p[1];
    const float keep = hi ? p[1] : p[0];
    return keep + wave_xor_dpp1(send);
}

// ---- the persistent kernels' per-step exchange, data-tagged (round 3: 10 -> 8.4 (fence-free counter) -> 5.6 us per step) ----
// A counter barrier costs four dependent round trips to the memory-side atomics per step (max, arrival, poll, key read).
// Here a workgroup PUBLISHES its candidate `b` in its own slot as two 8-byte {step, payload} granules (one relaxed
// agent-scope store each: the data is the flag, guide recipe R2) and one wave per workgroup SWEEPS all slots -- <= 256 x 2
// granules, eight 8-byte loads per lane, all in flight together -- until every tag equals the step, then takes the maximum
// itself: one store and (typically) two sweeps per step.  Slots are double-buffered by step parity: a workgroup can run at
// most one step ahead of the slowest (its next sweep needs everybody's next store), so a slot is never overwritten before
// every sweep of its previous use is over.  Called by one whole wave; returns the step's winner, `gave_up` set when the
// bounded wait ran out (or another workgroup raised status[1]).
__device__ __forceinline__ unsigned long long ps_exchange(unsigned long long b, int step, unsigned long long* __restrict__ gran,
                                                          unsigned int* __restrict__ status, int lane, unsigned int& gave_up) {
    unsigned long long* ga = gran + (size_t)(step & 1) * 2 * PS_MAXWG;           // [2][PS_MAXWG]: value granules, index granules
    const unsigned long long tag = (unsigned long long)(unsigned int)step << 32;
    if (lane == 0) {
        __hip_atomic_store(ga + blockIdx.x, tag | (b >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(ga + PS_MAXWG + blockIdx.x, tag | (b & 0xFFFFFFFFull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    unsigned long long win = 0ull;
    gave_up = 0;
    for (unsigned int polls = 0;; ++polls) {
        bool ok = true;
        win = 0ull;
#pragma unroll
        for (int k = 0; k < PS_MAXWG / 64; ++k) {
            const int slot = k * 64 + lane;
            if (slot < (int)gridDim.x) {
                const unsigned long long va = __hip_atomic_load(ga + slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned long long vi = __hip_atomic_load(ga + PS_MAXWG + slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = ok && (va >> 32) == (tag >> 32) && (vi >> 32) == (tag >> 32);
                const unsigned long long key = (va << 32) | (vi & 0xFFFFFFFFull);
                win = key > win ? key : win;
            }
        }
        if (__all(ok)) break;
        if ((polls & 15u) == 15u && __hip_atomic_load(&status[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { gave_up = 1; break; }
        if (polls > PS_SPIN_LIMIT) {
            if (lane == 0) __hip_atomic_store(&status[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            gave_up = 1;
            break;
        }
        __builtin_amdgcn_s_sleep(1);
    }
    win = wave_max_u64(win);
    return win;
}

template <int NG>
__global__ __launch_bounds__(PS_W * 64) void ms_seed_persistent_kernel(const float* __restrict__ X, int n,
                                                                     unsigned long long* __restrict__ keys, int num_seeds,
                                                                     unsigned int* __restrict__ status /* [1] abort */,
                                                                     unsigned long long* __restrict__ gran /* [2][2][PS_MAXWG] */) {
    __shared__ unsigned long long red[PS_W];
    __shared__ unsigned long long prev_s;
    __shared__ unsigned int abort_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, grp = lane >> 4;
    const int blk0 = blockIdx.x * (PS_W * 4 * 16 * NG);
    float4 x[NG][16];
    float near[NG];
    int rowj[NG];
#pragma unroll
    for (int t = 0; t < NG; ++t) {
        const int base = blk0 + ((wave * 4 + grp) * NG + t) * 16;
#pragma unroll
        for (int i = 0; i < 16; ++i)
            x[t][i] = *reinterpret_cast<const float4*>(X + (int64_t)min(base + i, n - 1) * MS_D + j * 4);
        rowj[t] = base + j;
        near[t] = INFINITY;
    }
    if (tid == 0) prev_s = __hip_atomic_load(&keys[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    for (int step = 1; step < num_seeds; ++step) {
        const unsigned int cur = 0xFFFFFFFFu - (unsigned int)(prev_s & 0xFFFFFFFFull);
        if (cur >= (unsigned int)n) return;      // only after an abort elsewhere (keys left at 0): leave, uniformly
        const float4 s = *reinterpret_cast<const float4*>(X + (int64_t)cur * MS_D + j * 4);
        unsigned long long best = 0ull;
#pragma unroll
        for (int t = 0; t < NG; ++t) {
            float p[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) p[i] = x[t][i].x * s.x + x[t][i].y * s.y + x[t][i].z * s.z + x[t][i].w * s.w;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const bool hi = j & 8;
                const float send = hi ? p[i] : p[i + 8];
                const float keep = hi ? p[i + 8] : p[i];
                p[i] = keep + wave_xor_dpp8(send);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const bool hi = j & 4;
                const float send = hi ? p[i] : p[i + 4];
                const float keep = hi ? p[i + 4] : p[i];
                p[i] = keep + wave_xor_dpp4(send);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const bool hi = j & 2;
                const float send = hi ? p[i] : p[i + 2];
                const float keep = hi ? p[i + 2] : p[i];
                p[i] = keep + wave_xor_dpp2(send);
            }
            {
                const bool hi = j & 1;
                const float send = hi ? p[0] : p[1];
                const float keep = hi ? p[1] : p[0];
                p[0] = keep + wave_xor_dpp1(send);
            }
            if (rowj[t] < n) {
                const float d = fminf(near[t], 0.5f * (1.0f - p[0]));
                near[t] = d;
                const unsigned long long key =
                    ((unsigned long long)ordered_bits(d) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned int)rowj[t]);
                best = key > best ? key : best;
            }
        }
        best = wave_max_u64(best);
        if (lane == 0) red[wave] = best;
        __syncthreads();
        if (wave == 0) {
            unsigned long long b = red[0];
#pragma unroll
            for (int w = 1; w < PS_W; ++w) b = red[w] > b ? red[w] : b;
            unsigned int gave_up;
            const unsigned long long win = ps_exchange(b, step, gran, status, lane, gave_up);     // see ps_exchange
            if (lane == 0) {
                abort_s = gave_up;
                prev_s = win;                                        // the winner of this step: the next step's row
                if (blockIdx.x == 0 && !gave_up) keys[step] = win;   // for the finish kernel (next launch)
            }
        }
        __syncthreads();
        if (abort_s != 0u) return;               // uniform per block after the barrier
    }
}

// ---- persistent seeding over the bf16 copy for maps BEYOND the register file (round 4) ---------------------------------------
// At 1280x960 the bf16 copy is 157 MB: no longer register-resident, but most of it still fits ON CHIP.  One launch, one
// workgroup per CU, and three homes for a workgroup's rows: NG tiles of 16 rows per 16-lane group in VGPRs (256 x 32 x NG x 16
// = 655 360 rows at NG = 5), NL tiles per group in LDS (262 144 rows at NL = 2, 132 KiB per CU; a group's tiles are offset by
// 128 B so the two groups of a 32-lane ds_read_b64 phase use opposite bank halves), and the remaining rows (311 296 = 40 MB at
// 1280x960) streamed from HBM every step with their nearest-distance in global memory, as the stepwise kernel does for
// all of them.  Every row's arithmetic is that of ms_seed_step_bf16_kernel (two v_dot2 per row chunk, the same butterfly,
// the same (value, ~index) key), so the selected indices are bit-identical to the stepwise bf16 path; the steps' exchange is
// ps_exchange.  Tiles past the end of the map re-cover its last 16 rows (duplicates cannot change a maximum).
constexpr int PB_GROUPS = PS_W * 4;                               // 16-lane groups per workgroup

template <int NG, int NL>
__global__ __launch_bounds__(PS_W * 64) void ms_seed_persistent_bf16_kernel(const uint16_t* __restrict__ Xb, int n,
                                                                          unsigned long long* __restrict__ keys, int num_seeds,
                                                                          unsigned int* __restrict__ status, unsigned long long* __restrict__ gran,
                                                                          float* __restrict__ nearest_tail, int tail0, int tail_rows_per_wg) {
    constexpr int LDS_GROUP = NL * 16 * 128 + 128;                // bytes of a group's LDS tiles (+128: alternate bank halves)
    extern __shared__ __attribute__((aligned(16))) unsigned char pb_lds[];
    __shared__ unsigned long long red[PS_W];
    __shared__ unsigned long long prev_s;
    __shared__ unsigned int abort_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, grp = lane >> 4, gi = wave * 4 + grp;
    const int gslot = blockIdx.x * PB_GROUPS + gi;
    const int reg_rows = gridDim.x * PB_GROUPS * NG * 16;
    uint2 x[NG][16];
    float near[NG], near_l[NL];
#pragma unroll
    for (int t = 0; t < NG; ++t) {
        const uint16_t* src = Xb + (int64_t)min((gslot * NG + t) * 16, n - 16) * MS_D + j * 4;
#pragma unroll
        for (int i = 0; i < 16; ++i) x[t][i] = *reinterpret_cast<const uint2*>(src + i * MS_D);
        near[t] = INFINITY;
    }
    unsigned char* ltile = pb_lds + gi * LDS_GROUP + j * 8;
#pragma unroll
    for (int t = 0; t < NL; ++t) {
        const uint16_t* src = Xb + (int64_t)min(reg_rows + (gslot * NL + t) * 16, n - 16) * MS_D + j * 4;
#pragma unroll
        for (int i = 0; i < 16; ++i) *reinterpret_cast<uint2*>(ltile + (t * 16 + i) * 128) = *reinterpret_cast<const uint2*>(src + i * MS_D);
        near_l[t] = INFINITY;
    }
    // the streamed tail of this workgroup: rows [t0, t1), 512 per pass
    const int t0 = min(n, tail0 + blockIdx.x * tail_rows_per_wg), t1 = min(n, t0 + tail_rows_per_wg);
    // The tail's first pass does not depend on the step's seed: its rows are requested before the previous step's exchange and
    // arrive behind it.
    uint2 xs[16];
    const bool has_tail = t0 + gi * 16 < t1;
    const uint16_t* tail_src = Xb + (int64_t)min(t0 + gi * 16, n - 16) * MS_D + j * 4;
    if (has_tail) {
#pragma unroll
        for (int i = 0; i < 16; ++i) xs[i] = *reinterpret_cast<const uint2*>(tail_src + i * MS_D);
    }
    if (tid == 0) prev_s = __hip_atomic_load(&keys[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    for (int step = 1; step < num_seeds; ++step) {
        const unsigned int cur = 0xFFFFFFFFu - (unsigned int)(prev_s & 0xFFFFFFFFull);
        if (cur >= (unsigned int)n) return;      // only after an abort elsewhere (keys left at 0): leave, uniformly
        const uint2 s = *reinterpret_cast<const uint2*>(Xb + (int64_t)cur * MS_D + j * 4);
        unsigned long long best = 0ull;
        // LDS tiles first, then the tail's first pass goes out and is in flight behind the register tiles
#pragma unroll
        for (int t = 0; t < NL; ++t) {
            float p[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) p[i] = dot4_bf16(*reinterpret_cast<const uint2*>(ltile + (t * 16 + i) * 128), s);
            const float d = fminf(near_l[t], 0.5f * (1.0f - butterfly16(p, j)));
            near_l[t] = d;
            const int row = min(reg_rows + (gslot * NL + t) * 16, n - 16) + j;
            const unsigned long long key = ((unsigned long long)ordered_bits(d) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned int)row);
            best = key > best ? key : best;
            __builtin_amdgcn_sched_barrier(0);       // one tile at a time: interleaved tiles cost 16 live partials each
        }
        int base = t0 + gi * 16, gbs = min(base, n - 16);
        float near_s = (step > 1 && base < t1) ? nearest_tail[gbs + j - tail0] : INFINITY;
#pragma unroll
        for (int t = 0; t < NG; ++t) {
            float p[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) p[i] = dot4_bf16(x[t][i], s);
            const float d = fminf(near[t], 0.5f * (1.0f - butterfly16(p, j)));
            near[t] = d;
            const int row = min((gslot * NG + t) * 16, n - 16) + j;
            const unsigned long long key = ((unsigned long long)ordered_bits(d) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned int)row);
            best = key > best ? key : best;
            __builtin_amdgcn_sched_barrier(0);
        }
        for (; base < t1; base += PB_GROUPS * 16) {
            float p[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) p[i] = dot4_bf16(xs[i], s);
            const int row = gbs + j;
            const float near_now = near_s;
            // the next pass's rows go out before this pass's butterfly (the dot products above were the last use of xs)
            const int nbase = base + PB_GROUPS * 16;
            if (nbase < t1) {
                gbs = min(nbase, n - 16);
                const uint16_t* src = Xb + (int64_t)gbs * MS_D + j * 4;
#pragma unroll
                for (int i = 0; i < 16; ++i) xs[i] = *reinterpret_cast<const uint2*>(src + i * MS_D);
                if (step > 1) near_s = nearest_tail[gbs + j - tail0];
            }
            const float d = fminf(near_now, 0.5f * (1.0f - butterfly16(p, j)));
            nearest_tail[row - tail0] = d;
            const unsigned long long key = ((unsigned long long)ordered_bits(d) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned int)row);
            best = key > best ? key : best;
        }
        if (has_tail && step + 1 < num_seeds) {
#pragma unroll
            for (int i = 0; i < 16; ++i) xs[i] = *reinterpret_cast<const uint2*>(tail_src + i * MS_D);
        }
        best = wave_max_u64(best);
        if (lane == 0) red[wave] = best;
        __syncthreads();
        if (wave == 0) {
            unsigned long long b = red[0];
#pragma unroll
            for (int w = 1; w < PS_W; ++w) b = red[w] > b ? red[w] : b;
            unsigned int gave_up;
            const unsigned long long win = ps_exchange(b, step, gran, status, lane, gave_up);
            if (lane == 0) {
                abort_s = gave_up;
                prev_s = win;
                if (blockIdx.x == 0 && !gave_up) keys[step] = win;
            }
        }
        __syncthreads();
        if (abort_s != 0u) return;
    }
}

__global__ void ms_seed_status_init_kernel(unsigned int* __restrict__ status, unsigned int give_up, unsigned long long* __restrict__ gran) {
    if (threadIdx.x < 2) status[threadIdx.x] = threadIdx.x == 1 ? give_up : 0u;
    for (int i = threadIdx.x; i < 4 * PS_MAXWG; i += blockDim.x) gran[i] = 0ull;     // tag 0 = no step: polled words are re-initialised every call
}

__global__ void ms_seed_init_kernel(unsigned long long* __restrict__ keys, int num_seeds, int64_t first) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < num_seeds) keys[i] = i == 0 ? (0xFFFFFFFF00000000ull | (unsigned long long)(0xFFFFFFFFu - (unsigned int)first)) : 0ull;
}

__global__ void ms_seed_finish_kernel(const float* __restrict__ X, const unsigned long long* __restrict__ keys,
                                      int64_t* __restrict__ sel, float* __restrict__ seeds, const unsigned int* __restrict__ status,
                                      int n) {
    const int i = blockIdx.x;
    unsigned int idx = 0xFFFFFFFFu - (unsigned int)(keys[i] & 0xFFFFFFFFull);
    if ((status && status[1] != 0u) || idx >= (unsigned int)n) {      // persistent kernel gave up: report, do not fabricate
        if (threadIdx.x == 0) sel[i] = -1;
        idx = 0;
    } else if (threadIdx.x == 0) sel[i] = (int64_t)idx;
    if (threadIdx.x < MS_D) seeds[(int64_t)i * MS_D + threadIdx.x] = X[(int64_t)idx * MS_D + threadIdx.x];
}

// ------------------------------------------------------------------------------------------------
// shared MFMA tile: scores of a 16-point block against all seeds.
//   X block staged in a per-wave LDS slab xs[16][SZ]; Z in zs[S16][SZ] (rows >= S are zero).
//   st[sb][r] = X[point lq*4 + r] . Z[seed sb*16 + lj]
// The 64-wide dot is walked as d = 16*(l>>4) + t so both fragments are LDS reads at stride 1.
// ------------------------------------------------------------------------------------------------
template <int NSB>
__device__ __forceinline__ void score_block(const float* __restrict__ xs, const float* __restrict__ zs, int lj, int lq,
                                            f32x4 (&st)[NSB]) {
    float4 xa[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) xa[u] = *reinterpret_cast<const float4*>(xs + lj * SZ + lq * 16 + u * 4);
#pragma unroll
    for (int sb = 0; sb < NSB; ++sb) {
        f32x4 a0 = f32x4{0.f, 0.f, 0.f, 0.f}, a1 = a0;      // two chains: no back-to-back dependent MFMAs
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float4 z = *reinterpret_cast<const float4*>(zs + (sb * 16 + lj) * SZ + lq * 16 + u * 4);
            a0 = mfma16(xa[u].x, z.x, a0);
            a1 = mfma16(xa[u].y, z.y, a1);
            a0 = mfma16(xa[u].z, z.z, a0);
            a1 = mfma16(xa[u].w, z.w, a1);
        }
        st[sb] = a0 + a1;
    }
}

__device__ __forceinline__ void stage_points(const float* __restrict__ X, int n, int p0, float* __restrict__ xs, int lane) {
    // 16 rows x 64 floats: 4 coalesced 1 KiB wave loads, 16-byte LDS stores
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = i * 4 + (lane >> 4), c4 = (lane & 15) * 4;
        const int src = min(p0 + row, n - 1);
        *reinterpret_cast<float4*>(xs + row * SZ + c4) = *reinterpret_cast<const float4*>(X + (int64_t)src * MS_D + c4);
    }
}

// ---- hill climbing: part[wg] = sum over the workgroup's points of exp(kappa s) x ----------------
template <int NSB>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void ms_hill_kernel(const float* __restrict__ X, int n, const float* __restrict__ Z,
                                                      int S, float kappa, float* __restrict__ part) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* zs = lds;                          // [NSB*16][SZ]
    float* xsa = lds + NSB * 16 * SZ;         // [4 waves][16][SZ]
    float* accum = lds;                       // [NSB*16][64] cross-wave reduction, reuses zs after the loop
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lj = lane & 15, lq = lane >> 4;
    for (int i = tid; i < NSB * 16 * MS_D; i += 256) {
        const int s = i / MS_D, d = i - s * MS_D;
        zs[s * SZ + d] = (s < S) ? Z[(int64_t)s * MS_D + d] : 0.f;
    }
    __syncthreads();
    float* xs = xsa + wave * 16 * SZ;

    f32x4 zn[NSB][4];
#pragma unroll
    for (int sb = 0; sb < NSB; ++sb)
#pragma unroll
        for (int db = 0; db < 4; ++db) zn[sb][db] = f32x4{0.f, 0.f, 0.f, 0.f};

    const float kl2 = kappa * 1.4426950408889634f;
    const int nblocks = (n + 15) / 16;
    // register-staged prefetch: the next 16 points are in flight while this slab's 32*NSB MFMAs run
    const int srow = lane >> 4, scol = (lane & 15) * 4;
#define MS_LOAD_SLAB(P0)                                                                                          \
    nx0 = *reinterpret_cast<const float4*>(X + (int64_t)min((P0) + srow, n - 1) * MS_D + scol);                    \
    nx1 = *reinterpret_cast<const float4*>(X + (int64_t)min((P0) + 4 + srow, n - 1) * MS_D + scol);                \
    nx2 = *reinterpret_cast<const float4*>(X + (int64_t)min((P0) + 8 + srow, n - 1) * MS_D + scol);                \
    nx3 = *reinterpret_cast<const float4*>(X + (int64_t)min((P0) + 12 + srow, n - 1) * MS_D + scol);
    float4 nx0, nx1, nx2, nx3;
    MS_LOAD_SLAB((blockIdx.x * 4 + wave) * 16)
    for (int pb = blockIdx.x * 4 + wave; pb < nblocks; pb += gridDim.x * 4) {
        const int p0 = pb * 16;
        *reinterpret_cast<float4*>(xs + srow * SZ + scol) = nx0;
        *reinterpret_cast<float4*>(xs + (4 + srow) * SZ + scol) = nx1;
        *reinterpret_cast<float4*>(xs + (8 + srow) * SZ + scol) = nx2;
        *reinterpret_cast<float4*>(xs + (12 + srow) * SZ + scol) = nx3;
        MS_LOAD_SLAB((pb + (int)gridDim.x * 4) * 16)
        // the slab is private to this wave; a wave-level fence orders the LDS writes before the reads
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        f32x4 st[NSB];
        score_block<NSB>(xs, zs, lj, lq, st);
        // B operand of W X: X[point lq*4 + r][d = db*16 + lj]
        float xb[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int db = 0; db < 4; ++db) xb[r][db] = xs[(lq * 4 + r) * SZ + db * 16 + lj];
        // exp(kappa*s) through v_exp_f32: |kappa*s*log2e| <= 29 at kappa = 20, relative error ~2e-6 (MS:26).  Only the last slab of
        // the point set can hold rows beyond n (clamped copies, weight 0): every other slab skips the per-row test.
        if (p0 + 16 <= n) {
#pragma unroll
            for (int sb = 0; sb < NSB; ++sb) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float w = __builtin_amdgcn_exp2f(kl2 * st[sb][r]);
#pragma unroll
                    for (int db = 0; db < 4; ++db) zn[sb][db] = mfma16(w, xb[r][db], zn[sb][db]);
                }
            }
        } else {
#pragma unroll
            for (int sb = 0; sb < NSB; ++sb) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float w = (p0 + lq * 4 + r < n) ? __builtin_amdgcn_exp2f(kl2 * st[sb][r]) : 0.f;
#pragma unroll
                    for (int db = 0; db < 4; ++db) zn[sb][db] = mfma16(w, xb[r][db], zn[sb][db]);
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    // deterministic cross-wave reduction: waves add into `accum` one after another
    __syncthreads();   // every wave is done reading zs
    for (int i = tid; i < NSB * 16 * MS_D; i += 256) accum[i] = 0.f;
    __syncthreads();
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int sb = 0; sb < NSB; ++sb)
#pragma unroll
                for (int db = 0; db < 4; ++db)
#pragma unroll
                    for (int r = 0; r < 4; ++r) accum[(sb * 16 + lq * 4 + r) * MS_D + db * 16 + lj] += zn[sb][db][r];
        }
        __syncthreads();
    }
    float* dst = part + (int64_t)blockIdx.x * (NSB * 16 * MS_D);
    for (int i = tid; i < NSB * 16 * MS_D; i += 256) dst[i] = accum[i];
}

#undef MS_LOAD_SLAB

// ---- the same step with fp32 results on the bf16 matrix pipe (msm_ms_hill_climb_split) -----------------------------
// Every fp32 operand is an exact sum of three bf16 terms (bf16.h: split3), a product keeps the six terms above 2^-24 of it
// and runs on v_mfma_f32_16x16x32_bf16: 6 x 16 cycles for 16x16x32 against 8 x 32 cycles of the fp32 instruction (0.375 of its
// matrix time).  The splitting is VALU work beside the MFMAs, and X is needed in two operand orders (k = channel for the
// scores, k = point for W X), so the kernel is arranged around what a wave can keep in registers:
//   * a wave owns a 32-point slab (wave-private fp32 tile in LDS, register-staged prefetch of the next one) and HALF of the
//     launch's seed blocks: waves w and w + 4 of the 512-thread workgroup walk the same slabs with seed blocks [0, NSBW) and
//     [NSBW, nsb) -- 16 NSBW accumulator registers instead of 16 nsb, at the price of splitting X twice;
//   * scores: A = X terms (point lj, channels 8 lq ..+7 of a 32-channel half), B = Z terms read from LDS planes that were
//     split once per launch; small terms and the h.h term in separate accumulators, added at the end;
//   * the D layout of the scores (lane = seed lj, registers = points 4 lq + r) IS the A layout of W X when the k index of the
//     K = 32 instruction is read as k = 8 kq + e <-> point (e < 4 ? 4 kq + e : 16 + 4 kq + e - 4): the eight exp() weights of
//     a lane (two 16-point score tiles) are split in place, the B operand gathers the same eight points of column 16 db + lj.
// One workgroup per CU (2 waves per SIMD: one wave's splitting runs beside the other's MFMAs).
constexpr int ZP_LD = MS_D + 8;        // bf16 row stride of the Z planes: 144 B = 9 slots of 16 B -> b128 reads of 8 rows hit 8 slots
constexpr int HS_ROWS = 32;            // points per slab

__device__ __forceinline__ f32x4 mfma_k32(const bf16x8& a, const bf16x8& b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }

template <int NSBW>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void ms_hill_split_kernel(const float* __restrict__ X, int n,
                                                                                                          const float* __restrict__ Z, int S, int nsb,
                                                                                                          float kappa, float* __restrict__ part) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int zrows = nsb * 16;
    uint16_t* zp = reinterpret_cast<uint16_t*>(lds);                               // [3][zrows][ZP_LD] bf16 terms of Z
    float* xsa = lds + (3 * zrows * ZP_LD) / 2;                                    // [8 waves][32][SZ]
    float* accum = lds;                                                            // [zrows][64] after the loop
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: scalar branches
    const int lj = lane & 15, lq = lane >> 4;
    for (int i = tid; i < zrows * (MS_D / 4); i += 512) {
        const int s = i / (MS_D / 4), c4 = (i - s * (MS_D / 4)) * 4;
        const float4 v = (s < S) ? *reinterpret_cast<const float4*>(Z + (int64_t)s * MS_D + c4) : make_float4(0.f, 0.f, 0.f, 0.f);
        const Split3 t = split3(v.x, v.y, v.z, v.w);
        *reinterpret_cast<u32x2b*>(zp + (0 * zrows + s) * ZP_LD + c4) = __builtin_bit_cast(u32x2b, t.h);
        *reinterpret_cast<u32x2b*>(zp + (1 * zrows + s) * ZP_LD + c4) = __builtin_bit_cast(u32x2b, t.m);
        *reinterpret_cast<u32x2b*>(zp + (2 * zrows + s) * ZP_LD + c4) = __builtin_bit_cast(u32x2b, t.l);
    }
    __syncthreads();
    const int pg = wave & 3, sb0 = (wave >> 2) * NSBW;
    const int nb = min(NSBW, nsb - sb0);                 // this wave's seed blocks (may be 0 when nsb == 1)
    float* xs = xsa + wave * HS_ROWS * SZ;

    f32x4 zn[NSBW][4];
#pragma unroll
    for (int sb = 0; sb < NSBW; ++sb)
#pragma unroll
        for (int db = 0; db < 4; ++db) zn[sb][db] = f32x4{0.f, 0.f, 0.f, 0.f};

    const float kl2 = kappa * 1.4426950408889634f;
    const int nslabs = (n + HS_ROWS - 1) / HS_ROWS;
    const int srow = lane >> 4, scol = (lane & 15) * 4;
    // register-staged prefetch of the next slab (eight named registers: an indexed array ends up in scratch)
#define HS_ROW(P0, I) (X + (int64_t)min((P0) + (I) * 4 + srow, n - 1) * MS_D + scol)
#define HS_LOAD_SLAB(P0)                                                                                              \
    nx0 = *reinterpret_cast<const float4*>(HS_ROW(P0, 0)); nx1 = *reinterpret_cast<const float4*>(HS_ROW(P0, 1));     \
    nx2 = *reinterpret_cast<const float4*>(HS_ROW(P0, 2)); nx3 = *reinterpret_cast<const float4*>(HS_ROW(P0, 3));     \
    nx4 = *reinterpret_cast<const float4*>(HS_ROW(P0, 4)); nx5 = *reinterpret_cast<const float4*>(HS_ROW(P0, 5));     \
    nx6 = *reinterpret_cast<const float4*>(HS_ROW(P0, 6)); nx7 = *reinterpret_cast<const float4*>(HS_ROW(P0, 7));
#define HS_STORE_ROW(I, V) *reinterpret_cast<float4*>(xs + ((I) * 4 + srow) * SZ + scol) = V;
    float4 nx0, nx1, nx2, nx3, nx4, nx5, nx6, nx7;
    HS_LOAD_SLAB((blockIdx.x * 4 + pg) * HS_ROWS)
    for (int sl = blockIdx.x * 4 + pg; sl < nslabs; sl += gridDim.x * 4) {
        const int p0 = sl * HS_ROWS;
        HS_STORE_ROW(0, nx0) HS_STORE_ROW(1, nx1) HS_STORE_ROW(2, nx2) HS_STORE_ROW(3, nx3)
        HS_STORE_ROW(4, nx4) HS_STORE_ROW(5, nx5) HS_STORE_ROW(6, nx6) HS_STORE_ROW(7, nx7)
        HS_LOAD_SLAB((sl + (int)gridDim.x * 4) * HS_ROWS)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // A operands of the scores: [16-point tile q][channel half h]
        Split3x8 xa[2][2];
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const float* src = xs + (q * 16 + lj) * SZ + h * 32 + lq * 8;
                const float4 v0 = *reinterpret_cast<const float4*>(src), v1 = *reinterpret_cast<const float4*>(src + 4);
                xa[q][h] = join(split3(v0.x, v0.y, v0.z, v0.w), split3(v1.x, v1.y, v1.z, v1.w));
            }
        // B operands of W X: the lane's eight points of column 16 db + lj
        Split3x8 xb[4];
#pragma unroll
        for (int db = 0; db < 4; ++db) {
            float e[8];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                e[r] = xs[(lq * 4 + r) * SZ + db * 16 + lj];
                e[4 + r] = xs[(16 + lq * 4 + r) * SZ + db * 16 + lj];
            }
            xb[db] = join(split3(e[0], e[1], e[2], e[3]), split3(e[4], e[5], e[6], e[7]));
        }
        const bool tail = p0 + HS_ROWS > n;     // only the last slab can hold rows beyond n (clamped copies, weight 0)
        // Software pipeline over the seed blocks: the score MFMAs of block sb + 1 are issued before the exp() / split work of
        // block sb, so that a wave's own vector work runs beside its own matrix work as well as beside the other wave's.
        f32x4 sa = f32x4{0.f, 0.f, 0.f, 0.f}, sbv = sa;
        auto score = [&](int sb, f32x4& oa, f32x4& ob) {
            bf16x8 zf[2][3];
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int t = 0; t < 3; ++t)
                    zf[h][t] = *reinterpret_cast<const bf16x8*>(zp + (t * zrows + (sb0 + sb) * 16 + lj) * ZP_LD + h * 32 + lq * 8);
            f32x4 alo = f32x4{0.f, 0.f, 0.f, 0.f}, ahi = alo, blo = alo, bhi = alo;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                alo = mfma_k32(xa[0][h].l, zf[h][0], alo);
                blo = mfma_k32(xa[1][h].l, zf[h][0], blo);
                ahi = mfma_k32(xa[0][h].h, zf[h][0], ahi);
                bhi = mfma_k32(xa[1][h].h, zf[h][0], bhi);
                alo = mfma_k32(xa[0][h].h, zf[h][2], alo);
                blo = mfma_k32(xa[1][h].h, zf[h][2], blo);
                alo = mfma_k32(xa[0][h].m, zf[h][1], alo);
                blo = mfma_k32(xa[1][h].m, zf[h][1], blo);
                alo = mfma_k32(xa[0][h].m, zf[h][0], alo);
                blo = mfma_k32(xa[1][h].m, zf[h][0], blo);
                alo = mfma_k32(xa[0][h].h, zf[h][1], alo);
                blo = mfma_k32(xa[1][h].h, zf[h][1], blo);
            }
            oa = alo + ahi;
            ob = blo + bhi;
        };
        if (nb > 0) score(0, sa, sbv);
#pragma unroll
        for (int sb = 0; sb < NSBW; ++sb) {
            if (sb < nb) {
                const f32x4 ca = sa, cb = sbv;
                if (sb + 1 < NSBW && sb + 1 < nb) score(sb + 1, sa, sbv);
                float w[8];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    w[r] = __builtin_amdgcn_exp2f(kl2 * ca[r]);
                    w[4 + r] = __builtin_amdgcn_exp2f(kl2 * cb[r]);
                }
                if (tail) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (p0 + lq * 4 + r >= n) w[r] = 0.f;
                        if (p0 + 16 + lq * 4 + r >= n) w[4 + r] = 0.f;
                    }
                }
                const Split3x8 w3 = join(split3(w[0], w[1], w[2], w[3]), split3(w[4], w[5], w[6], w[7]));
                // six terms, smallest first, four independent chains (one per column block)
#pragma unroll
                for (int db = 0; db < 4; ++db) zn[sb][db] = mfma_k32(w3.l, xb[db].h, zn[sb][db]);
#pragma unroll
                for (int db = 0; db < 4; ++db) zn[sb][db] = mfma_k32(w3.h, xb[db].l, zn[sb][db]);
#pragma unroll
                for (int db = 0; db < 4; ++db) zn[sb][db] = mfma_k32(w3.m, xb[db].m, zn[sb][db]);
#pragma unroll
                for (int db = 0; db < 4; ++db) zn[sb][db] = mfma_k32(w3.m, xb[db].h, zn[sb][db]);
#pragma unroll
                for (int db = 0; db < 4; ++db) zn[sb][db] = mfma_k32(w3.h, xb[db].m, zn[sb][db]);
#pragma unroll
                for (int db = 0; db < 4; ++db) zn[sb][db] = mfma_k32(w3.h, xb[db].h, zn[sb][db]);
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
#undef HS_LOAD_SLAB
#undef HS_STORE_ROW
#undef HS_ROW
    // deterministic reduction over the four point groups: waves pg = 0..3 add into `accum` in turn (the two waves of a turn
    // own disjoint seed rows)
    __syncthreads();   // every wave is done reading the Z planes
    for (int i = tid; i < zrows * MS_D; i += 512) accum[i] = 0.f;
    __syncthreads();
    for (int g = 0; g < 4; ++g) {
        if (pg == g) {
#pragma unroll
            for (int sb = 0; sb < NSBW; ++sb)
                if (sb < nb) {
#pragma unroll
                    for (int db = 0; db < 4; ++db)
#pragma unroll
                        for (int r = 0; r < 4; ++r) accum[((sb0 + sb) * 16 + lq * 4 + r) * MS_D + db * 16 + lj] += zn[sb][db][r];
                }
        }
        __syncthreads();
    }
    float* dst = part + (int64_t)blockIdx.x * (zrows * MS_D);
    for (int i = tid; i < zrows * MS_D; i += 512) dst[i] = accum[i];
}

// ---- the split form with X split ONCE per call (default of msm_ms_hill_climb_split) -------------------------------------------
// ms_hill_split_kernel above is bound by vector issue: 2/3 of its non-matrix instructions split X, and X does not change over
// the iterations.  Here a pre-pass writes X as three bf16 planes (ms_split_planes_kernel, 6 bytes per element instead of 4
// read per iteration) and the iteration kernel touches X with no vector instruction at all:
//   * a slab's three 32 x 64 bf16 planes (12 KiB) travel HBM -> LDS by LDS-DMA (global_load_lds_dwordx4), double-buffered per
//     wave pair, the 16-byte chunks XOR-swizzled through the SOURCE address (chunk c of row r sits at slot c ^ (r & 7): the
//     b128 reads of 8 rows hit 8 different slots);
//   * score A operands are ds_read_b128 of a row's 8 channels; the W X B operands (8 POINTS of one channel per lane) come from
//     the same row-major tile through ds_read_b64_tr_b16, which hands lane c of a 16-lane group column c of the 4 x 16 block the
//     group's lanes address (lane i: row i / 4, elements 4 (i % 4) .. + 3);
//   * what is left on the vector pipe is exp() and the split of the weights: ~90 instructions per 48 MFMAs.
// The two waves of a pair share the tile and hand it over through an LDS counter.
constexpr int HP_PLANE = HS_ROWS * MS_D * 2;       // bytes of one plane of a slab tile
constexpr int HP_TILE = 3 * HP_PLANE;

__global__ __launch_bounds__(256) void ms_split_planes_kernel(const float* __restrict__ X, int n, int n_pad, uint16_t* __restrict__ planes) {
    const int64_t total4 = (int64_t)n_pad * (MS_D / 4), ps = (int64_t)n_pad * MS_D;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (int64_t)gridDim.x * 256) {
        const float4 v = (i < (int64_t)n * (MS_D / 4)) ? *reinterpret_cast<const float4*>(X + i * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        const Split3 t = split3(v.x, v.y, v.z, v.w);
        *reinterpret_cast<u32x2b*>(planes + i * 4) = __builtin_bit_cast(u32x2b, t.h);
        *reinterpret_cast<u32x2b*>(planes + ps + i * 4) = __builtin_bit_cast(u32x2b, t.m);
        *reinterpret_cast<u32x2b*>(planes + 2 * ps + i * 4) = __builtin_bit_cast(u32x2b, t.l);
    }
}

// lane l's 16 bytes at sbase + voff(l) land at LDS byte address lds_dst + 16 l (see enc_block.hip: glds16)
__device__ __forceinline__ void ms_glds16(const void* sbase, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(lds_dst)
                 : "memory");
}
typedef short v4i16_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bf16x4 lds_read_tr16(const char* p) {
    const v4i16_t r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4i16_t __attribute__((address_space(3)))*)(uintptr_t)(unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)p);
    return __builtin_bit_cast(bf16x4, r);
}

template <int NSBW>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void ms_hill_planes_kernel(const uint16_t* __restrict__ Xp, int64_t plane_stride,
                                                                                                           int n, const float* __restrict__ Z, int S,
                                                                                                           int nsb, float kappa, float* __restrict__ part) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int zrows = nsb * 16;
    uint16_t* zp = reinterpret_cast<uint16_t*>(lds);                               // [3][zrows][ZP_LD] bf16 terms of Z
    char* tiles = reinterpret_cast<char*>(lds) + (size_t)3 * zrows * ZP_LD * 2;    // [4 pairs][2 buffers][HP_TILE]
    float* accum = lds;                                                            // [zrows][64] after the loop
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lj = lane & 15, lq = lane >> 4;
    for (int i = tid; i < zrows * (MS_D / 4); i += 512) {
        const int s = i / (MS_D / 4), c4 = (i - s * (MS_D / 4)) * 4;
        const float4 v = (s < S) ? *reinterpret_cast<const float4*>(Z + (int64_t)s * MS_D + c4) : make_float4(0.f, 0.f, 0.f, 0.f);
        const Split3 t = split3(v.x, v.y, v.z, v.w);
        *reinterpret_cast<u32x2b*>(zp + (0 * zrows + s) * ZP_LD + c4) = __builtin_bit_cast(u32x2b, t.h);
        *reinterpret_cast<u32x2b*>(zp + (1 * zrows + s) * ZP_LD + c4) = __builtin_bit_cast(u32x2b, t.m);
        *reinterpret_cast<u32x2b*>(zp + (2 * zrows + s) * ZP_LD + c4) = __builtin_bit_cast(u32x2b, t.l);
    }
    const int pg = wave & 3, sh = wave >> 2, sb0 = sh * NSBW;
    const int nb = min(NSBW, nsb - sb0);
    char* tile0 = tiles + pg * 2 * HP_TILE;
    const unsigned tile_lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)tile0;

    f32x4 zn[NSBW][4];
#pragma unroll
    for (int sb = 0; sb < NSBW; ++sb)
#pragma unroll
        for (int db = 0; db < 4; ++db) zn[sb][db] = f32x4{0.f, 0.f, 0.f, 0.f};

    const float kl2 = kappa * 1.4426950408889634f;
    const int nslabs = (n + HS_ROWS - 1) / HS_ROWS;
    const int stride = (int)gridDim.x * 4;
    const int iters = (nslabs - (int)blockIdx.x * 4 + stride - 1) / stride;      // of the workgroup's first pair: the others run as many
    // DMA piece I = 0..11 of a tile: plane I / 4, rows 8 (I % 4) .. + 7; lane L writes slot L of the piece = row L / 8, position L % 8,
    // which holds chunk (L % 8) ^ (row & 7) of that row
    const unsigned dma_off = (unsigned)((lane >> 3) * 128 + (((lane & 7) ^ ((lane >> 3) & 7)) * 16));
    auto issue = [&](int sl, int buf) {
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const int I = sh * 6 + k, plane = I >> 2, quarter = I & 3;
            const char* sbase = reinterpret_cast<const char*>(Xp + (int64_t)plane * plane_stride + ((int64_t)sl * HS_ROWS + quarter * 8) * MS_D);
            ms_glds16(sbase, dma_off, tile_lds + (unsigned)(buf * HP_TILE + plane * HP_PLANE + quarter * 1024));
        }
    };
    // per-lane byte offsets into a plane of the tile
    unsigned a_off[2], b_off[4];
#pragma unroll
    for (int h = 0; h < 2; ++h) a_off[h] = (unsigned)((lj * 8 + ((h * 4 + lq) ^ (lj & 7))) * 16);            // row lj (+16 q), channels 32 h + 8 lq ..
    {
        const int row = 4 * lq + (lj >> 2);                                                                  // (+16 for the second half of k)
#pragma unroll
        for (int db = 0; db < 4; ++db) b_off[db] = (unsigned)((row * 8 + ((db * 2 + ((lj & 3) >> 1)) ^ (row & 7))) * 16 + (lj & 1) * 8);
    }
    const int sl0 = (int)blockIdx.x * 4 + pg;
    if (sl0 < nslabs) issue(sl0, 0);
    // The two waves of a pair meet at an LDS counter, not at a workgroup barrier (the pairs drift apart instead of reading their
    // operands all at once: 566 -> 556 us): a wave adds 1 when its DMA pieces of tile `it` have landed -- which it only waits for
    // after its reads of the other buffer -- and goes on when the count says both did.
    __shared__ int arrive[4];
    if (tid < 4) arrive[tid] = 0;
    __syncthreads();              // Z planes staged, counters cleared
    for (int it = 0; it < iters; ++it) {
        const int sl = sl0 + it * stride, buf = it & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_fetch_add(&arrive[pg], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        while (__hip_atomic_load(&arrive[pg], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < 2 * (it + 1)) __builtin_amdgcn_s_sleep(1);
        if (sl + stride < nslabs) issue(sl + stride, buf ^ 1);
        if (sl < nslabs) {
            const int p0 = sl * HS_ROWS;
            const char* T = tile0 + buf * HP_TILE;
            Split3x8 xa[2][2];
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const char* src = T + a_off[h] + q * 2048;
                    xa[q][h].h = *reinterpret_cast<const bf16x8*>(src);
                    xa[q][h].m = *reinterpret_cast<const bf16x8*>(src + HP_PLANE);
                    xa[q][h].l = *reinterpret_cast<const bf16x8*>(src + 2 * HP_PLANE);
                }
            Split3x8 xb[4];
#pragma unroll
            for (int db = 0; db < 4; ++db) {
                const char* src = T + b_off[db];
                xb[db].h = cat8(lds_read_tr16(src), lds_read_tr16(src + 2048));
                xb[db].m = cat8(lds_read_tr16(src + HP_PLANE), lds_read_tr16(src + HP_PLANE + 2048));
                xb[db].l = cat8(lds_read_tr16(src + 2 * HP_PLANE), lds_read_tr16(src + 2 * HP_PLANE + 2048));
            }
            const bool tail = p0 + HS_ROWS > n;
            f32x4 sa = f32x4{0.f, 0.f, 0.f, 0.f}, sbv = sa;
            auto score = [&](int sb, f32x4& oa, f32x4& ob) {
                bf16x8 zf[2][3];
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int t = 0; t < 3; ++t)
                        zf[h][t] = *reinterpret_cast<const bf16x8*>(zp + (t * zrows + (sb0 + sb) * 16 + lj) * ZP_LD + h * 32 + lq * 8);
                f32x4 alo = f32x4{0.f, 0.f, 0.f, 0.f}, ahi = alo, blo = alo, bhi = alo;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    alo = mfma_k32(xa[0][h].l, zf[h][0], alo);
                    blo = mfma_k32(xa[1][h].l, zf[h][0], blo);
                    ahi = mfma_k32(xa[0][h].h, zf[h][0], ahi);
                    bhi = mfma_k32(xa[1][h].h, zf[h][0], bhi);
                    alo = mfma_k32(xa[0][h].h, zf[h][2], alo);
                    blo = mfma_k32(xa[1][h].h, zf[h][2], blo);
                    alo = mfma_k32(xa[0][h].m, zf[h][1], alo);
                    blo = mfma_k32(xa[1][h].m, zf[h][1], blo);
                    alo = mfma_k32(xa[0][h].m, zf[h][0], alo);
                    blo = mfma_k32(xa[1][h].m, zf[h][0], blo);
                    alo = mfma_k32(xa[0][h].h, zf[h][1], alo);
                    blo = mfma_k32(xa[1][h].h, zf[h][1], blo);
                }
                oa = alo + ahi;
                ob = blo + bhi;
            };
            if (nb > 0) score(0, sa, sbv);
#pragma unroll
            for (int sb = 0; sb < NSBW; ++sb) {
                if (sb < nb) {
                    const f32x4 ca = sa, cb = sbv;
                    if (sb + 1 < NSBW && sb + 1 < nb) score(sb + 1, sa, sbv);
                    float w[8];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        w[r] = __builtin_amdgcn_exp2f(kl2 * ca[r]);
                        w[4 + r] = __builtin_amdgcn_exp2f(kl2 * cb[r]);
                    }
                    if (tail) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            if (p0 + lq * 4 + r >= n) w[r] = 0.f;
                            if (p0 + 16 + lq * 4 + r >= n) w[4 + r] = 0.f;
                        }
                    }
                    const Split3x8 w3 = join(split3(w[0], w[1], w[2], w[3]), split3(w[4], w[5], w[6], w[7]));
#pragma unroll
                    for (int db = 0; db < 4; ++db) zn[sb][db] = mfma_k32(w3.l, xb[db].h, zn[sb][db]);
#pragma unroll
                    for (int db = 0; db < 4; ++db) zn[sb][db] = mfma_k32(w3.h, xb[db].l, zn[sb][db]);
#pragma unroll
                    for (int db = 0; db < 4; ++db) zn[sb][db] = mfma_k32(w3.m, xb[db].m, zn[sb][db]);
#pragma unroll
                    for (int db = 0; db < 4; ++db) zn[sb][db] = mfma_k32(w3.m, xb[db].h, zn[sb][db]);
#pragma unroll
                    for (int db = 0; db < 4; ++db) zn[sb][db] = mfma_k32(w3.h, xb[db].m, zn[sb][db]);
#pragma unroll
                    for (int db = 0; db < 4; ++db) zn[sb][db] = mfma_k32(w3.h, xb[db].h, zn[sb][db]);
                }
            }
        }
    }
    __syncthreads();   // every wave is done reading the Z planes
    for (int i = tid; i < zrows * MS_D; i += 512) accum[i] = 0.f;
    __syncthreads();
    for (int g = 0; g < 4; ++g) {
        if (pg == g) {
#pragma unroll
            for (int sb = 0; sb < NSBW; ++sb)
                if (sb < nb) {
#pragma unroll
                    for (int db = 0; db < 4; ++db)
#pragma unroll
                        for (int r = 0; r < 4; ++r) accum[((sb0 + sb) * 16 + lq * 4 + r) * MS_D + db * 16 + lj] += zn[sb][db][r];
                }
        }
        __syncthreads();
    }
    float* dst = part + (int64_t)blockIdx.x * (zrows * MS_D);
    for (int i = tid; i < zrows * MS_D; i += 512) dst[i] = accum[i];
}

// ---- the step in the low-precision mode (precision "bf16": BASELINE configs[4]) ---------------------------------------------------
// ms_hill_planes_kernel with ONE bf16 plane of X and single-term products: scores = Z(h + l) . x (the seeds keep both terms: an
// error in z moves every weight of its row the same way), W = bf16(exp(kappa s)), W X one MFMA per column block -- 12 MFMAs per
// (32-point slab, 16-seed block) instead of 48, ~25 vector instructions instead of ~90, 4 KiB of tile instead of 12.  With the
// operands a third of the size a wave carries up to ten seed blocks (160 accumulator registers), so all 19 blocks of 300 seeds
// are ONE launch per iteration and X (157 MB at n = 1 228 800) is read once per iteration instead of three times.
constexpr int HB_TILE = HS_ROWS * MS_D * 2;        // bytes of a slab tile (one bf16 plane)

template <int NSBW>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void ms_hill_bf16_kernel(const uint16_t* __restrict__ Xb, int n,
                                                                                                         const float* __restrict__ Z, int S, int nsb,
                                                                                                         float kappa, float* __restrict__ part) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int zrows = nsb * 16;
    uint16_t* zp = reinterpret_cast<uint16_t*>(lds);                               // [2][zrows][ZP_LD]: h, l terms of Z
    const size_t zbytes = (size_t)2 * zrows * ZP_LD * 2, abytes = (size_t)zrows * MS_D * 4;
    char* tiles = reinterpret_cast<char*>(lds) + (zbytes > abytes ? zbytes : abytes);      // [4 pairs][2 buffers][HB_TILE], clear of `accum`
    float* accum = lds;                                                            // [zrows][64] after the loop
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lj = lane & 15, lq = lane >> 4;
    for (int i = tid; i < zrows * (MS_D / 4); i += 512) {
        const int s = i / (MS_D / 4), c4 = (i - s * (MS_D / 4)) * 4;
        const float4 v = (s < S) ? *reinterpret_cast<const float4*>(Z + (int64_t)s * MS_D + c4) : make_float4(0.f, 0.f, 0.f, 0.f);
        const Split4 t = split4(v.x, v.y, v.z, v.w);
        *reinterpret_cast<u32x2b*>(zp + (0 * zrows + s) * ZP_LD + c4) = __builtin_bit_cast(u32x2b, t.hi);
        *reinterpret_cast<u32x2b*>(zp + (1 * zrows + s) * ZP_LD + c4) = __builtin_bit_cast(u32x2b, t.lo);
    }
    const int pg = wave & 3, sh = wave >> 2, sb0 = sh * NSBW;
    const int nb = min(NSBW, nsb - sb0);
    char* tile0 = tiles + pg * 2 * HB_TILE;
    const unsigned tile_lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)tile0;

    f32x4 zn[NSBW][4];
#pragma unroll
    for (int sb = 0; sb < NSBW; ++sb)
#pragma unroll
        for (int db = 0; db < 4; ++db) zn[sb][db] = f32x4{0.f, 0.f, 0.f, 0.f};

    const float kl2 = kappa * 1.4426950408889634f;
    const int nslabs = (n + HS_ROWS - 1) / HS_ROWS;
    const int stride = (int)gridDim.x * 4;
    const int iters = (nslabs - (int)blockIdx.x * 4 + stride - 1) / stride;
    // DMA piece I = 0..3 of a tile: rows 8 I .. + 7; lane L writes slot L of the piece = row L / 8, position L % 8, which holds
    // chunk (L % 8) ^ (row & 7) of that row (the swizzle of ms_hill_planes_kernel); the two waves of a pair take two pieces each
    const unsigned dma_off = (unsigned)((lane >> 3) * 128 + (((lane & 7) ^ ((lane >> 3) & 7)) * 16));
    auto issue = [&](int sl, int buf) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int quarter = sh * 2 + k;
            const char* sbase = reinterpret_cast<const char*>(Xb + ((int64_t)sl * HS_ROWS + quarter * 8) * MS_D);
            ms_glds16(sbase, dma_off, tile_lds + (unsigned)(buf * HB_TILE + quarter * 1024));
        }
    };
    unsigned a_off[2], b_off[4];
#pragma unroll
    for (int h = 0; h < 2; ++h) a_off[h] = (unsigned)((lj * 8 + ((h * 4 + lq) ^ (lj & 7))) * 16);
    {
        const int row = 4 * lq + (lj >> 2);
#pragma unroll
        for (int db = 0; db < 4; ++db) b_off[db] = (unsigned)((row * 8 + ((db * 2 + ((lj & 3) >> 1)) ^ (row & 7))) * 16 + (lj & 1) * 8);
    }
    const int sl0 = (int)blockIdx.x * 4 + pg;
    if (sl0 < nslabs) issue(sl0, 0);
    __shared__ int arrive_b[4];
    if (tid < 4) arrive_b[tid] = 0;
    __syncthreads();              // Z planes staged, counters cleared
    for (int it = 0; it < iters; ++it) {
        const int sl = sl0 + it * stride, buf = it & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_fetch_add(&arrive_b[pg], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        while (__hip_atomic_load(&arrive_b[pg], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < 2 * (it + 1)) __builtin_amdgcn_s_sleep(1);
        if (sl + stride < nslabs) issue(sl + stride, buf ^ 1);
        if (sl < nslabs) {
            const int p0 = sl * HS_ROWS;
            const char* T = tile0 + buf * HB_TILE;
            bf16x8 xa[2][2];
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int h = 0; h < 2; ++h) xa[q][h] = *reinterpret_cast<const bf16x8*>(T + a_off[h] + q * 2048);
            bf16x8 xb[4];
#pragma unroll
            for (int db = 0; db < 4; ++db) xb[db] = cat8(lds_read_tr16(T + b_off[db]), lds_read_tr16(T + b_off[db] + 2048));
            const bool tail = p0 + HS_ROWS > n;
            f32x4 sa = f32x4{0.f, 0.f, 0.f, 0.f}, sbv = sa;
            auto score = [&](int sb, f32x4& oa, f32x4& ob) {
                f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f}, b = a;
#pragma unroll
                for (int t = 1; t >= 0; --t)                    // low-order term first
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const bf16x8 zf = *reinterpret_cast<const bf16x8*>(zp + (t * zrows + (sb0 + sb) * 16 + lj) * ZP_LD + h * 32 + lq * 8);
                        a = mfma_k32(xa[0][h], zf, a);
                        b = mfma_k32(xa[1][h], zf, b);
                    }
                oa = a;
                ob = b;
            };
            if (nb > 0) score(0, sa, sbv);
#pragma unroll
            for (int sb = 0; sb < NSBW; ++sb) {
                if (sb < nb) {
                    const f32x4 ca = sa, cb = sbv;
                    if (sb + 1 < NSBW && sb + 1 < nb) score(sb + 1, sa, sbv);
                    float w[8];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        w[r] = __builtin_amdgcn_exp2f(kl2 * ca[r]);
                        w[4 + r] = __builtin_amdgcn_exp2f(kl2 * cb[r]);
                    }
                    if (tail) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            if (p0 + lq * 4 + r >= n) w[r] = 0.f;
                            if (p0 + 16 + lq * 4 + r >= n) w[4 + r] = 0.f;
                        }
                    }
                    const bf16x8 w8 = cat8(pack4(w[0], w[1], w[2], w[3]), pack4(w[4], w[5], w[6], w[7]));
#pragma unroll
                    for (int db = 0; db < 4; ++db) zn[sb][db] = mfma_k32(w8, xb[db], zn[sb][db]);
                }
            }
        }
    }
    __syncthreads();   // every wave is done reading the Z planes
    for (int i = tid; i < zrows * MS_D; i += 512) accum[i] = 0.f;
    __syncthreads();
    for (int g = 0; g < 4; ++g) {
        if (pg == g) {
#pragma unroll
            for (int sb = 0; sb < NSBW; ++sb)
                if (sb < nb) {
#pragma unroll
                    for (int db = 0; db < 4; ++db)
#pragma unroll
                        for (int r = 0; r < 4; ++r) accum[((sb0 + sb) * 16 + lq * 4 + r) * MS_D + db * 16 + lj] += zn[sb][db][r];
                }
        }
        __syncthreads();
    }
    float* dst = part + (int64_t)blockIdx.x * (zrows * MS_D);
    for (int i = tid; i < zrows * MS_D; i += 512) dst[i] = accum[i];
}

// Z[s] = normalize(sum_wg part[wg][s])  (MS:103 F.normalize).  16 waves per seed: wave w adds its fixed slice of
// the workgroup partials (8 loads in flight), then the slices are added in wave order -- deterministic.
__global__ __launch_bounds__(1024) void ms_hill_finish_kernel(const float* __restrict__ part, int nwg, int rows_padded,
                                                              float* __restrict__ Z) {
    __shared__ float slice[16][MS_D];
    const int s = blockIdx.x, d = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int per = (nwg + 15) / 16;
    const int g0 = w * per, g1 = min(nwg, g0 + per);
    const float* src = part + (int64_t)s * MS_D + d;
    const int64_t stride = (int64_t)rows_padded * MS_D;
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int g = g0;
    for (; g + 8 <= g1; g += 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u) a[u] += src[(int64_t)(g + u) * stride];
    }
    for (; g < g1; ++g) a[0] += src[(int64_t)g * stride];
    slice[w][d] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    __syncthreads();
    if (w == 0) {
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc += slice[i][d];
        const float nrm = fmaxf(sqrtf(wave_sum(acc * acc)), 1e-12f);
        Z[(int64_t)s * MS_D + d] = acc / nrm;
    }
}

// ---- connected components of the converged seeds ------------------------------------------------------------------
// mean_shift.py:41-76 is sequential and order dependent -- the i-th still-unlabelled seed claims every seed within epsilon
// (cosine distance 0.5 (1 - z_j . z_i)); if some of those already carry labels it takes their mode (smallest label on
// equal counts: np.unique sorts, argmax takes the first), otherwise a fresh label -- but every step of it is a parallel
// operation over <= 304 seeds.  One wave walks the sequence: the seeds sit in LDS, a step's S dot products are one per
// lane and chunk, the mode is an LDS histogram + a wave reduction.  What this buys is not the arithmetic (a dozen steps of
// a microsecond) but the HOST: the loop used to run there on a copy of the seeds, i.e. a device synchronisation and a
// transfer in the middle of every clustering, with the GPU idle behind it.
constexpr int CC_MAXS = MS_SB * 16;
__global__ __launch_bounds__(64) void ms_components_kernel(const float* __restrict__ Z, int S, float eps, int64_t* __restrict__ labels_out,
                                                           int32_t* __restrict__ num_out) {
    extern __shared__ __attribute__((aligned(16))) float cz[];           // [S][MS_D + 1], then int lab[S], cnt[S]
    constexpr int ZS = MS_D + 1;
    int* lab = reinterpret_cast<int*>(cz + (size_t)S * ZS);
    int* cnt = lab + S;
    const int lane = threadIdx.x;
    for (int i = lane; i < S * MS_D; i += 64) cz[(i / MS_D) * ZS + (i % MS_D)] = Z[i];
    for (int j = lane; j < S; j += 64) lab[j] = -1;
    __syncthreads();
    int K = 0;
    constexpr int NCH = (CC_MAXS + 63) / 64;
    for (int i = 0; i < S; ++i) {
        if (lab[i] != -1) continue;                                      // uniform
        for (int l = lane; l < K; l += 64) cnt[l] = 0;
        __syncthreads();
        bool comp[NCH];
        bool labelled = false;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int j = c * 64 + lane;
            comp[c] = false;
            if (j < S) {
                float dot = 0.f;
#pragma unroll 16
                for (int k = 0; k < MS_D; ++k) dot = fmaf(cz[j * ZS + k], cz[i * ZS + k], dot);
                comp[c] = 0.5f * (1.0f - dot) <= eps;
                if (comp[c] && lab[j] >= 0) {
                    atomicAdd(&cnt[lab[j]], 1);
                    labelled = true;
                }
            }
        }
        __syncthreads();
        int label;
        if (__any(labelled)) {
            // mode of the labels already present (MS:30-38, 66-68): largest count, smallest label on ties
            int bc = -1, bl = 0x7fffffff;
            for (int l = lane; l < K; l += 64)
                if (cnt[l] > bc) { bc = cnt[l]; bl = l; }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const int oc = __shfl_xor(bc, o, 64), ol = __shfl_xor(bl, o, 64);
                if (oc > bc || (oc == bc && ol < bl)) { bc = oc; bl = ol; }
            }
            label = bl;
        } else {
            label = K++;
        }
#pragma unroll
        for (int c = 0; c < NCH; ++c)
            if (comp[c]) lab[c * 64 + lane] = label;
        __syncthreads();
    }
    for (int j = lane; j < S; j += 64) labels_out[j] = (int64_t)lab[j];
    // num_out[0] = labels that SURVIVE (a later step may overwrite every seed of an earlier label: the reference's
    // `num = len(unique(seed_labels))`, MS:211, then counts only labels 0 .. num - 1), num_out[1] = labels created
    for (int l = lane; l < K; l += 64) cnt[l] = 0;
    __syncthreads();
    for (int j = lane; j < S; j += 64) cnt[lab[j]] = 1;
    __syncthreads();
    int alive = 0;
    for (int l = lane; l < K; l += 64) alive += cnt[l];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) alive += __shfl_xor(alive, o, 64);
    if (lane == 0) {
        num_out[0] = alive;
        num_out[1] = K;
    }
}

// ---- assignment -----------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ms_assign_kernel(const float* __restrict__ X, int n, const float* __restrict__ Z, int S,
                                                        int nchunks, const int64_t* __restrict__ seed_labels,
                                                        int64_t* __restrict__ labels_out,
                                                        unsigned long long* __restrict__ counts, int num_labels) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int rows = nchunks * MS_CH * 16;
    float* zs = lds;                                                             // [rows][SZ]
    float* xsa = lds + rows * SZ;                                                // [4][16][SZ]
    unsigned int* hist = reinterpret_cast<unsigned int*>(xsa + 4 * 16 * SZ);   // [num_labels]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lj = lane & 15, lq = lane >> 4;
    for (int i = tid; i < rows * MS_D; i += 256) {
        const int s = i / MS_D, d = i - s * MS_D;
        zs[s * SZ + d] = (s < S) ? Z[(int64_t)s * MS_D + d] : 0.f;
    }
    for (int i = tid; i < num_labels; i += 256) hist[i] = 0u;
    __syncthreads();
    float* xs = xsa + wave * 16 * SZ;
    const int nblocks = (n + 15) / 16;
    for (int pb = blockIdx.x * 4 + wave; pb < nblocks; pb += gridDim.x * 4) {
        const int p0 = pb * 16;
        stage_points(X, n, p0, xs, lane);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // first argmin over seeds of 0.5*(1 - dot) (MS:206-209): seeds grow with chunk, sb, then lj
        float bd[4] = {INFINITY, INFINITY, INFINITY, INFINITY};
        int bi[4] = {0x7fffffff, 0x7fffffff, 0x7fffffff, 0x7fffffff};
        for (int c = 0; c < nchunks; ++c) {
            f32x4 st[MS_CH];
            score_block<MS_CH>(xs, zs + c * MS_CH * 16 * SZ, lj, lq, st);
#pragma unroll
            for (int sb = 0; sb < MS_CH; ++sb) {
                const int seed = (c * MS_CH + sb) * 16 + lj;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float dist = 0.5f * (1.0f - st[sb][r]);
                    if (seed < S && dist < bd[r]) { bd[r] = dist; bi[r] = seed; }
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) {
                const float od = __shfl_xor(bd[r], o, 64);
                const int oi = __shfl_xor(bi[r], o, 64);
                if (od < bd[r] || (od == bd[r] && oi < bi[r])) { bd[r] = od; bi[r] = oi; }
            }
            const int p = p0 + lq * 4 + r;
            if (lj == 0 && p < n) {
                const int64_t lab = seed_labels[min(bi[r], S - 1)];        // all-NaN distances leave bi at its sentinel
                labels_out[p] = lab;
                if (lab >= 0 && lab < num_labels) atomicAdd(&hist[(int)lab], 1u);
            }
        }
    }
    __syncthreads();
    for (int i = tid; i < num_labels; i += 256)
        if (hist[i]) atomicAdd(&counts[i], (unsigned long long)hist[i]);
}

__global__ __launch_bounds__(256) void ms_relabel_kernel(int64_t* __restrict__ labels, int n, const int64_t* __restrict__ counts,
                                                         int num_labels, const int32_t* __restrict__ num_alive) {
    // first argmax of counts (torch.argmax, MS:222) over labels 0 .. num - 1, num = number of distinct seed labels (MS:211-216:
    // with a vanished label the label values have gaps and the reference never counts the ones >= num); every thread
    // recomputes it (<= 304 entries)
    if (num_alive != nullptr) num_labels = min(num_labels, max((int)num_alive[0], 1));
    int lmax = 0;
    int64_t best = counts[0];
    for (int i = 1; i < num_labels; ++i) {
        const int64_t c = counts[i];
        if (c > best) { best = c; lmax = i; }
    }
    if (lmax == 0) return;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t l = labels[i];
        if (l == 0) labels[i] = lmax;
        else if (l == lmax) labels[i] = 0;
    }
}

// whole 256-row passes per workgroup (every lane group busy), at most ~2048 atomics on the step's key
static int seed_blocks(int n) {
    const int passes = max(1, cdiv(n, 256 * 2048));
    return max(1, cdiv(n, 256 * passes));
}
// seed blocks per hill-climb launch: equal chunks of at most `cap` blocks (MSM_OPT_MS_CHUNK overrides the cap)
static int hill_chunk(int nsb) {
    const int v = opt(MSM_OPT_MS_CHUNK);
    const int cap = (v >= 1 && v <= MS_CH) ? v : MS_CH;
    return cdiv(nsb, cdiv(nsb, cap));
}
static int hill_wgs(int n) { return max(1, min(512, ((n + 15) / 16 + 3) / 4)); }

}  // namespace msm

using namespace msm;

// keys [S] u64 | 8 words (2 used: barrier arrivals, abort flag) | nearest [n] (stepwise path)
extern "C" int64_t msm_ms_seed_workspace(int n) { return (int64_t)n + 2 * (MS_SB * 16) + 16; }

extern "C" int msm_ms_select_seeds(const float* X, int n, int d, int num_seeds, int64_t first_index, float* seeds_out,
                                   int64_t* indices_out, float* workspace, int64_t workspace_elems, int flags, void* stream) {
    MSM_REQUIRE(X && seeds_out && indices_out && workspace, "msm_ms_select_seeds: null pointer");
    MSM_REQUIRE(d == MS_D, "msm_ms_select_seeds: d=%d, only d=64 is supported", d);
    MSM_REQUIRE(n > 0 && num_seeds > 0 && first_index >= 0 && first_index < n, "msm_ms_select_seeds: bad sizes");
    MSM_REQUIRE((((uintptr_t)X) & 15) == 0 && (((uintptr_t)workspace) & 7) == 0, "msm_ms_select_seeds: misaligned pointer");
    if (workspace_elems < msm_ms_seed_workspace(n)) {
        set_error("msm_ms_select_seeds: workspace too small");
        return MSM_E_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    MSM_REQUIRE(num_seeds <= MS_SB * 16, "msm_ms_select_seeds: at most %d seeds", MS_SB * 16);
    const int nblk = seed_blocks(n);
    unsigned long long* keys = reinterpret_cast<unsigned long long*>(workspace);       // [num_seeds]
    float* nearest = workspace + 2 * (MS_SB * 16) + 8;
    hipLaunchKernelGGL(ms_seed_init_kernel, dim3(cdiv(num_seeds, 64)), dim3(64), 0, st, keys, num_seeds, first_index);
    // persistent single-launch path when the map fits the register files of the CUs (see ms_seed_persistent_kernel)
    // CU count of the CURRENT device (a process may drive several): cached per device ordinal
    static int cu_cache[64] = {0};
    int dev = 0, n_cus = 0;
    if (hipGetDevice(&dev) == hipSuccess) {
        if (dev >= 0 && dev < 64 && cu_cache[dev] > 0) n_cus = cu_cache[dev];
        else if (hipDeviceGetAttribute(&n_cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess) {
            if (dev >= 0 && dev < 64) cu_cache[dev] = n_cus;
        } else n_cus = 0;
    }
    const int ng = cdiv(n, 256 * PS_W * 64);                       // rows per workgroup = 512 * ng
    const int pgrid = ng >= 1 && ng <= 3 ? cdiv(n, PS_W * 64 * ng) : 0;
    unsigned int* status = reinterpret_cast<unsigned int*>(workspace + 2 * (MS_SB * 16) + 4);   // 2 words between keys and nearest
    if (pgrid > 0 && pgrid <= n_cus && pgrid <= PS_MAXWG && n >= 4096 && num_seeds > 2 && !(flags & MSM_MS_SEED_STEPWISE) &&
        opt(MSM_OPT_MS_NO_PERSISTENT) != 1) {
        // the exchange slots live where the stepwise path keeps nearest[] (unused here): 2 parities x 2 granule rows x 256 slots x 8 B = 8 KiB
        unsigned long long* gran = reinterpret_cast<unsigned long long*>(nearest);
        hipLaunchKernelGGL(ms_seed_status_init_kernel, dim3(1), dim3(256), 0, st, status, (flags & MSM_MS_SEED_TEST_GIVE_UP) ? 1u : 0u, gran);
        switch (ng) {
            case 1: hipLaunchKernelGGL(ms_seed_persistent_kernel<1>, dim3(pgrid), dim3(PS_W * 64), 0, st, X, n, keys, num_seeds, status, gran); break;
            case 2: hipLaunchKernelGGL(ms_seed_persistent_kernel<2>, dim3(pgrid), dim3(PS_W * 64), 0, st, X, n, keys, num_seeds, status, gran); break;
            default: hipLaunchKernelGGL(ms_seed_persistent_kernel<3>, dim3(pgrid), dim3(PS_W * 64), 0, st, X, n, keys, num_seeds, status, gran); break;
        }
        hipLaunchKernelGGL(ms_seed_finish_kernel, dim3(num_seeds), dim3(64), 0, st, X, keys, indices_out, seeds_out, status, n);
        MSM_CHECK_LAUNCH("msm_ms_select_seeds(persistent)");
        return MSM_OK;
    }
    for (int i = 1; i < num_seeds; ++i)
        if (n >= 16) hipLaunchKernelGGL(ms_seed_step_kernel<false>, dim3(nblk), dim3(256), 0, st, X, n, keys, i, nearest);
        else hipLaunchKernelGGL(ms_seed_step_kernel<true>, dim3(nblk), dim3(256), 0, st, X, n, keys, i, nearest);
    hipLaunchKernelGGL(ms_seed_finish_kernel, dim3(num_seeds), dim3(64), 0, st, X, keys, indices_out, seeds_out,
                       (const unsigned int*)nullptr, n);
    MSM_CHECK_LAUNCH("msm_ms_select_seeds");
    return MSM_OK;
}

extern "C" int64_t msm_ms_hill_climb_workspace(int n, int S) {
    const int nsb = cdiv(S, 16);
    return (int64_t)hill_wgs(n) * nsb * 16 * MS_D;
}

template <int NSB>
static int hill_chunk_launch(const float* X, int n, const float* Zc, int Sc, float kappa, float* ws, int G, hipStream_t st) {
    const size_t lds = sizeof(float) * ((size_t)NSB * 16 * SZ + 4 * 16 * SZ);
    MSM_CHECK_HIP((hipError_t)ensure_dynamic_lds((const void*)ms_hill_kernel<NSB>, lds));
    hipLaunchKernelGGL((ms_hill_kernel<NSB>), dim3(G), dim3(256), lds, st, X, n, Zc, Sc, kappa, ws);
    return MSM_OK;
}

extern "C" int msm_ms_hill_climb(const float* X, int n, int d, float* Z, int S, float kappa, int iters, float* workspace,
                                 int64_t workspace_elems, void* stream) {
    MSM_REQUIRE(X && Z && workspace, "msm_ms_hill_climb: null pointer");
    MSM_REQUIRE(d == MS_D, "msm_ms_hill_climb: d=%d, only d=64 is supported", d);
    MSM_REQUIRE(n > 0 && S > 0 && S <= MS_SB * 16 && iters >= 0, "msm_ms_hill_climb: bad sizes (S <= %d)", MS_SB * 16);
    MSM_REQUIRE((((uintptr_t)X) & 15) == 0, "msm_ms_hill_climb: X must be 16-byte aligned");
    if (workspace_elems < msm_ms_hill_climb_workspace(n, S)) {
        set_error("msm_ms_hill_climb: workspace too small");
        return MSM_E_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    const int G = hill_wgs(n);
    const int nsb = cdiv(S, 16);
    const int CH = hill_chunk(nsb);
    for (int it = 0; it < iters; ++it) {
        // all chunks of one iteration read the same Z; the finish kernels run after every chunk
        float* ws = workspace;
        for (int b0 = 0; b0 < nsb; b0 += CH) {
            const int nb = min(CH, nsb - b0);
            const float* Zc = Z + (int64_t)b0 * 16 * MS_D;
            const int Sc = min(S - b0 * 16, nb * 16);
            int rc = MSM_OK;
            switch (nb) {
                case 1: rc = hill_chunk_launch<1>(X, n, Zc, Sc, kappa, ws, G, st); break;
                case 2: rc = hill_chunk_launch<2>(X, n, Zc, Sc, kappa, ws, G, st); break;
                case 3: rc = hill_chunk_launch<3>(X, n, Zc, Sc, kappa, ws, G, st); break;
                case 4: rc = hill_chunk_launch<4>(X, n, Zc, Sc, kappa, ws, G, st); break;
                case 5: rc = hill_chunk_launch<5>(X, n, Zc, Sc, kappa, ws, G, st); break;
                case 6: rc = hill_chunk_launch<6>(X, n, Zc, Sc, kappa, ws, G, st); break;
                case 7: rc = hill_chunk_launch<7>(X, n, Zc, Sc, kappa, ws, G, st); break;
                default: rc = hill_chunk_launch<8>(X, n, Zc, Sc, kappa, ws, G, st); break;
            }
            if (rc != MSM_OK) return rc;
            ws += (int64_t)G * nb * 16 * MS_D;
        }
        ws = workspace;
        for (int b0 = 0; b0 < nsb; b0 += CH) {
            const int nb = min(CH, nsb - b0);
            const int Sc = min(S - b0 * 16, nb * 16);
            hipLaunchKernelGGL(ms_hill_finish_kernel, dim3(Sc), dim3(1024), 0, st, ws, G, nb * 16, Z + (int64_t)b0 * 16 * MS_D);
            ws += (int64_t)G * nb * 16 * MS_D;
        }
    }
    MSM_CHECK_LAUNCH("msm_ms_hill_climb");
    return MSM_OK;
}

template <int NSBW>
static int hill_split_launch(const float* X, int n, const float* Zc, int Sc, int nb, float kappa, float* ws, int G, hipStream_t st) {
    const size_t lds = (size_t)3 * nb * 16 * ZP_LD * sizeof(uint16_t) + sizeof(float) * (size_t)8 * HS_ROWS * SZ;
    MSM_CHECK_HIP((hipError_t)ensure_dynamic_lds((const void*)ms_hill_split_kernel<NSBW>, lds));
    hipLaunchKernelGGL((ms_hill_split_kernel<NSBW>), dim3(G), dim3(512), lds, st, X, n, Zc, Sc, nb, kappa, ws);
    return MSM_OK;
}

template <int NSBW>
static int hill_planes_launch(const uint16_t* Xp, int64_t plane_stride, int n, const float* Zc, int Sc, int nb, float kappa, float* ws, int G,
                              hipStream_t st) {
    const size_t lds = (size_t)3 * nb * 16 * ZP_LD * sizeof(uint16_t) + (size_t)4 * 2 * HP_TILE;
    MSM_CHECK_HIP((hipError_t)ensure_dynamic_lds((const void*)ms_hill_planes_kernel<NSBW>, lds));
    hipLaunchKernelGGL((ms_hill_planes_kernel<NSBW>), dim3(G), dim3(512), lds, st, Xp, plane_stride, n, Zc, Sc, nb, kappa, ws);
    return MSM_OK;
}

// partial sums of the workgroups (as msm_ms_hill_climb) + the three bf16 planes of X, rows padded to whole 32-point slabs
extern "C" int64_t msm_ms_hill_climb_split_workspace(int n, int S) {
    const int64_t n_pad = (int64_t)cdiv(n, HS_ROWS) * HS_ROWS;
    return msm_ms_hill_climb_workspace(n, S) + 3 * n_pad * MS_D / 2 + 4;
}

extern "C" int msm_ms_hill_climb_split(const float* X, int n, int d, float* Z, int S, float kappa, int iters, float* workspace,
                                       int64_t workspace_elems, void* stream) {
    MSM_REQUIRE(X && Z && workspace, "msm_ms_hill_climb_split: null pointer");
    MSM_REQUIRE(d == MS_D, "msm_ms_hill_climb_split: d=%d, only d=64 is supported", d);
    MSM_REQUIRE(n > 0 && S > 0 && S <= MS_SB * 16 && iters >= 0, "msm_ms_hill_climb_split: bad sizes (S <= %d)", MS_SB * 16);
    MSM_REQUIRE((((uintptr_t)X) & 15) == 0 && (((uintptr_t)Z) & 15) == 0 && (((uintptr_t)workspace) & 15) == 0,
                "msm_ms_hill_climb_split: X, Z and workspace must be 16-byte aligned");
    if (workspace_elems < msm_ms_hill_climb_split_workspace(n, S)) {
        set_error("msm_ms_hill_climb_split: workspace too small");
        return MSM_E_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    // one 512-thread workgroup per CU, four 32-point slabs in flight per workgroup (never more workgroups than hill_wgs(n):
    // the partial-sum region is sized for those)
    const int nslabs = cdiv(n, HS_ROWS);
    const int G = max(1, min(256, cdiv(nslabs, 4)));
    const int nsb = cdiv(S, 16);
    const int CH = hill_chunk(nsb);
    const bool planes = opt(MSM_OPT_MS_SPLIT_KERNEL) != 1;         // 1: X split inside the iteration kernel (fallback, no pre-pass)
    const int64_t n_pad = (int64_t)nslabs * HS_ROWS;
    uint16_t* Xp = reinterpret_cast<uint16_t*>(workspace);
    float* parts = workspace + (3 * n_pad * MS_D / 2 + 3) / 4 * 4;
    if (planes && iters > 0)
        hipLaunchKernelGGL(ms_split_planes_kernel, dim3((unsigned)min((int64_t)2048, (n_pad * (MS_D / 4) + 255) / 256)), dim3(256), 0, st, X, n,
                           (int)n_pad, Xp);
    for (int it = 0; it < iters; ++it) {
        float* ws = parts;
        for (int b0 = 0; b0 < nsb; b0 += CH) {
            const int nb = min(CH, nsb - b0);
            const float* Zc = Z + (int64_t)b0 * 16 * MS_D;
            const int Sc = min(S - b0 * 16, nb * 16);
            int rc = MSM_OK;
            if (planes) {
                switch ((nb + 1) / 2) {
                    case 1: rc = hill_planes_launch<1>(Xp, n_pad * MS_D, n, Zc, Sc, nb, kappa, ws, G, st); break;
                    case 2: rc = hill_planes_launch<2>(Xp, n_pad * MS_D, n, Zc, Sc, nb, kappa, ws, G, st); break;
                    case 3: rc = hill_planes_launch<3>(Xp, n_pad * MS_D, n, Zc, Sc, nb, kappa, ws, G, st); break;
                    default: rc = hill_planes_launch<4>(Xp, n_pad * MS_D, n, Zc, Sc, nb, kappa, ws, G, st); break;
                }
            } else {
                switch ((nb + 1) / 2) {
                    case 1: rc = hill_split_launch<1>(X, n, Zc, Sc, nb, kappa, ws, G, st); break;
                    case 2: rc = hill_split_launch<2>(X, n, Zc, Sc, nb, kappa, ws, G, st); break;
                    case 3: rc = hill_split_launch<3>(X, n, Zc, Sc, nb, kappa, ws, G, st); break;
                    default: rc = hill_split_launch<4>(X, n, Zc, Sc, nb, kappa, ws, G, st); break;
                }
            }
            if (rc != MSM_OK) return rc;
            ws += (int64_t)G * nb * 16 * MS_D;
        }
        ws = parts;
        for (int b0 = 0; b0 < nsb; b0 += CH) {
            const int nb = min(CH, nsb - b0);
            const int Sc = min(S - b0 * 16, nb * 16);
            hipLaunchKernelGGL(ms_hill_finish_kernel, dim3(Sc), dim3(1024), 0, st, ws, G, nb * 16, Z + (int64_t)b0 * 16 * MS_D);
            ws += (int64_t)G * nb * 16 * MS_D;
        }
    }
    MSM_CHECK_LAUNCH("msm_ms_hill_climb_split");
    return MSM_OK;
}

// ---- precision "bf16": one bf16 copy of X shared by seeding and the hill climb -------------------------------------------------------
extern "C" int64_t msm_ms_bf16_rows(int n) { return (int64_t)cdiv(n, HS_ROWS) * HS_ROWS; }

extern "C" int msm_ms_pack_bf16(const float* X, int n, int d, void* Xb, void* stream) {
    MSM_REQUIRE(X && Xb, "msm_ms_pack_bf16: null pointer");
    MSM_REQUIRE(d == MS_D && n > 0, "msm_ms_pack_bf16: d=%d (only 64), n=%d", d, n);
    MSM_REQUIRE(((((uintptr_t)X) | ((uintptr_t)Xb)) & 15) == 0, "msm_ms_pack_bf16: pointers must be 16-byte aligned");
    const int64_t n_pad = msm_ms_bf16_rows(n);
    hipLaunchKernelGGL(ms_pack_bf16_kernel, dim3((unsigned)min((int64_t)4096, (n_pad * (MS_D / 4) + 255) / 256)), dim3(256), 0, (hipStream_t)stream, X, n,
                       (int)n_pad, (uint16_t*)Xb);
    MSM_CHECK_LAUNCH("msm_ms_pack_bf16");
    return MSM_OK;
}

extern "C" int msm_ms_select_seeds_bf16(const void* Xb, const float* X, int n, int d, int num_seeds, int64_t first_index, float* seeds_out,
                                        int64_t* indices_out, float* workspace, int64_t workspace_elems, int flags, void* stream) {
    MSM_REQUIRE(Xb && X && seeds_out && indices_out && workspace, "msm_ms_select_seeds_bf16: null pointer");
    MSM_REQUIRE(d == MS_D, "msm_ms_select_seeds_bf16: d=%d, only d=64 is supported", d);
    MSM_REQUIRE(n >= 16 && num_seeds > 0 && first_index >= 0 && first_index < n, "msm_ms_select_seeds_bf16: bad sizes (n >= 16)");
    MSM_REQUIRE(num_seeds <= MS_SB * 16, "msm_ms_select_seeds_bf16: at most %d seeds", MS_SB * 16);
    MSM_REQUIRE((((uintptr_t)Xb) & 15) == 0 && (((uintptr_t)workspace) & 7) == 0, "msm_ms_select_seeds_bf16: misaligned pointer");
    if (workspace_elems < msm_ms_seed_workspace(n)) {
        set_error("msm_ms_select_seeds_bf16: workspace too small");
        return MSM_E_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    unsigned long long* keys = reinterpret_cast<unsigned long long*>(workspace);
    float* nearest = workspace + 2 * (MS_SB * 16) + 8;
    hipLaunchKernelGGL(ms_seed_init_kernel, dim3(cdiv(num_seeds, 64)), dim3(64), 0, st, keys, num_seeds, first_index);
    // one persistent launch with the rows held in VGPRs / LDS / streamed (ms_seed_persistent_bf16_kernel): one workgroup per CU
    // CU count of the CURRENT device (a process may drive several): cached per device ordinal
    static int cu_cache[64] = {0};
    int dev = 0, n_cus = 0;
    if (hipGetDevice(&dev) == hipSuccess) {
        if (dev >= 0 && dev < 64 && cu_cache[dev] > 0) n_cus = cu_cache[dev];
        else if (hipDeviceGetAttribute(&n_cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess) {
            if (dev >= 0 && dev < 64) cu_cache[dev] = n_cus;
        } else n_cus = 0;
    }
    constexpr int NG = 5, NL = 2;
    const int pgrid = min(n_cus, PS_MAXWG);
    const int tail0 = pgrid * PB_GROUPS * (NG + NL) * 16;
    if (pgrid >= 64 && n >= 65536 && (n <= tail0 || n - tail0 >= 16) && num_seeds > 2 && !(flags & MSM_MS_SEED_STEPWISE) &&
        opt(MSM_OPT_MS_NO_PERSISTENT) != 1) {
        unsigned int* status = reinterpret_cast<unsigned int*>(workspace + 2 * (MS_SB * 16) + 4);
        unsigned long long* gran = reinterpret_cast<unsigned long long*>(nearest);          // 8 KiB of exchange slots, then the tail's nearest[]
        float* nearest_tail = nearest + 4 * PS_MAXWG * 2;
        const int tail_rows_per_wg = n > tail0 ? cdiv(cdiv(n - tail0, pgrid), 16) * 16 : 0;
        const size_t lds = (size_t)PB_GROUPS * (NL * 16 * 128 + 128);
        MSM_CHECK_HIP((hipError_t)ensure_dynamic_lds((const void*)ms_seed_persistent_bf16_kernel<NG, NL>, lds));
        hipLaunchKernelGGL(ms_seed_status_init_kernel, dim3(1), dim3(256), 0, st, status, (flags & MSM_MS_SEED_TEST_GIVE_UP) ? 1u : 0u, gran);
        hipLaunchKernelGGL((ms_seed_persistent_bf16_kernel<NG, NL>), dim3(pgrid), dim3(PS_W * 64), lds, st, (const uint16_t*)Xb, n, keys, num_seeds,
                           status, gran, nearest_tail, tail0, tail_rows_per_wg);
        hipLaunchKernelGGL(ms_seed_finish_kernel, dim3(num_seeds), dim3(64), 0, st, X, keys, indices_out, seeds_out, status, n);
        MSM_CHECK_LAUNCH("msm_ms_select_seeds_bf16(persistent)");
        return MSM_OK;
    }
    const int nblk = seed_blocks(n);
    for (int i = 1; i < num_seeds; ++i)
        hipLaunchKernelGGL(ms_seed_step_bf16_kernel, dim3(nblk), dim3(256), 0, st, (const uint16_t*)Xb, n, keys, i, nearest);
    // the seeds handed on are rows of the caller's fp32 X (the reference returns X[selected], MS:186-189)
    hipLaunchKernelGGL(ms_seed_finish_kernel, dim3(num_seeds), dim3(64), 0, st, X, keys, indices_out, seeds_out, (const unsigned int*)nullptr, n);
    MSM_CHECK_LAUNCH("msm_ms_select_seeds_bf16");
    return MSM_OK;
}

template <int NSBW>
static int hill_bf16_launch(const uint16_t* Xb, int n, const float* Zc, int Sc, int nb, float kappa, float* ws, int G, hipStream_t st) {
    const size_t zbytes = (size_t)2 * nb * 16 * ZP_LD * sizeof(uint16_t), abytes = (size_t)nb * 16 * MS_D * sizeof(float);
    const size_t lds = (zbytes > abytes ? zbytes : abytes) + (size_t)4 * 2 * HB_TILE;
    MSM_CHECK_HIP((hipError_t)ensure_dynamic_lds((const void*)ms_hill_bf16_kernel<NSBW>, lds));
    hipLaunchKernelGGL((ms_hill_bf16_kernel<NSBW>), dim3(G), dim3(512), lds, st, Xb, n, Zc, Sc, nb, kappa, ws);
    return MSM_OK;
}

extern "C" int msm_ms_hill_climb_bf16(const void* Xb, int n, int d, float* Z, int S, float kappa, int iters, float* workspace,
                                      int64_t workspace_elems, void* stream) {
    MSM_REQUIRE(Xb && Z && workspace, "msm_ms_hill_climb_bf16: null pointer");
    MSM_REQUIRE(d == MS_D, "msm_ms_hill_climb_bf16: d=%d, only d=64 is supported", d);
    MSM_REQUIRE(n > 0 && S > 0 && S <= MS_SB * 16 && iters >= 0, "msm_ms_hill_climb_bf16: bad sizes (S <= %d)", MS_SB * 16);
    MSM_REQUIRE((((uintptr_t)Xb) & 15) == 0 && (((uintptr_t)Z) & 15) == 0 && (((uintptr_t)workspace) & 15) == 0,
                "msm_ms_hill_climb_bf16: Xb, Z and workspace must be 16-byte aligned");
    if (workspace_elems < msm_ms_hill_climb_workspace(n, S)) {
        set_error("msm_ms_hill_climb_bf16: workspace too small");
        return MSM_E_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    const int nslabs = cdiv(n, HS_ROWS);
    const int G = max(1, min(256, cdiv(nslabs, 4)));
    const int nsb = cdiv(S, 16);
    const int nsbw = (nsb + 1) / 2;                                  // seed blocks per wave: all of them in one launch
    for (int it = 0; it < iters; ++it) {
        int rc = MSM_OK;
        switch (nsbw) {
            case 1: rc = hill_bf16_launch<1>((const uint16_t*)Xb, n, Z, S, nsb, kappa, workspace, G, st); break;
            case 2: rc = hill_bf16_launch<2>((const uint16_t*)Xb, n, Z, S, nsb, kappa, workspace, G, st); break;
            case 3: rc = hill_bf16_launch<3>((const uint16_t*)Xb, n, Z, S, nsb, kappa, workspace, G, st); break;
            case 4: rc = hill_bf16_launch<4>((const uint16_t*)Xb, n, Z, S, nsb, kappa, workspace, G, st); break;
            case 5: case 6: rc = hill_bf16_launch<6>((const uint16_t*)Xb, n, Z, S, nsb, kappa, workspace, G, st); break;
            case 7: case 8: rc = hill_bf16_launch<8>((const uint16_t*)Xb, n, Z, S, nsb, kappa, workspace, G, st); break;
            default: rc = hill_bf16_launch<10>((const uint16_t*)Xb, n, Z, S, nsb, kappa, workspace, G, st); break;
        }
        if (rc != MSM_OK) return rc;
        hipLaunchKernelGGL(ms_hill_finish_kernel, dim3(S), dim3(1024), 0, st, workspace, G, nsb * 16, Z);
    }
    MSM_CHECK_LAUNCH("msm_ms_hill_climb_bf16");
    return MSM_OK;
}

extern "C" int msm_ms_assign(const float* X, int n, int d, const float* Z, int S, const int64_t* seed_labels,
                             int64_t* labels_out, int64_t* counts, int num_labels, void* stream) {
    MSM_REQUIRE(X && Z && seed_labels && labels_out && counts, "msm_ms_assign: null pointer");
    MSM_REQUIRE(d == MS_D, "msm_ms_assign: d=%d, only d=64 is supported", d);
    MSM_REQUIRE(n > 0 && S > 0 && S <= MS_SB * 16 && num_labels > 0 && num_labels <= 4096, "msm_ms_assign: bad sizes");
    hipStream_t st = (hipStream_t)stream;
    MSM_CHECK_HIP(hipMemsetAsync(counts, 0, sizeof(int64_t) * (size_t)num_labels, st));
    const int nchunks = cdiv(cdiv(S, 16), MS_CH);
    const int G = hill_wgs(n);
    const size_t lds = sizeof(float) * ((size_t)nchunks * MS_CH * 16 * SZ + 4 * 16 * SZ) + sizeof(unsigned int) * (size_t)num_labels;
    MSM_CHECK_HIP((hipError_t)ensure_dynamic_lds((const void*)ms_assign_kernel, lds));
    hipLaunchKernelGGL(ms_assign_kernel, dim3(G), dim3(256), lds, st, X, n, Z, S, nchunks, seed_labels, labels_out,
                       reinterpret_cast<unsigned long long*>(counts), num_labels);
    MSM_CHECK_LAUNCH("msm_ms_assign");
    return MSM_OK;
}

extern "C" int msm_ms_connected_components(const float* Z, int S, int d, float epsilon, int64_t* seed_labels, int32_t* num_labels,
                                           void* stream) {
    MSM_REQUIRE(Z && seed_labels && num_labels, "msm_ms_connected_components: null pointer");
    MSM_REQUIRE(d == MS_D, "msm_ms_connected_components: d=%d, only d=64 is supported", d);
    MSM_REQUIRE(S > 0 && S <= CC_MAXS, "msm_ms_connected_components: S=%d must be in 1..%d", S, CC_MAXS);
    const size_t lds = sizeof(float) * (size_t)S * (MS_D + 1) + sizeof(int) * 2 * (size_t)S;
    MSM_CHECK_HIP((hipError_t)ensure_dynamic_lds((const void*)ms_components_kernel, lds));
    hipLaunchKernelGGL(ms_components_kernel, dim3(1), dim3(64), lds, (hipStream_t)stream, Z, S, epsilon, seed_labels, num_labels);
    MSM_CHECK_LAUNCH("msm_ms_connected_components");
    return MSM_OK;
}

extern "C" int msm_ms_relabel_largest_zero(int64_t* labels, int n, const int64_t* counts, int num_labels, const int32_t* num_alive,
                                           void* stream) {
    MSM_REQUIRE(labels && counts && n > 0 && num_labels > 0, "msm_ms_relabel_largest_zero: bad arguments");
    hipLaunchKernelGGL(ms_relabel_kernel, dim3(min(2048, cdiv(n, 256))), dim3(256), 0, (hipStream_t)stream, labels, n,
                       counts, num_labels, num_alive);
    MSM_CHECK_LAUNCH("msm_ms_relabel_largest_zero");
    return MSM_OK;
}
