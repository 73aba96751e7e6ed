// fp32 attention of libttl_hip_strict.so — the TEST-ONLY strict-precision build (common.hpp, -DTTL_OPERAND_FP32).
//
// Same launchers as attention.hip (kernels.hpp): o = softmax(q k^T / 8) v with the row logsumexp saved (HF modeling_clip.py:259-277),
// its backward with P recomputed from the logsumexp, and the two pooled-query forms of the last layer.  Everything in fp32 FMAs, a
// two-pass softmax (exact row maximum first), one thread per query (forward, dQ) or per key (dK / dV) with the other side of the head
// in LDS: written to be obviously right, not fast.  Never benched; the product builds do not contain this file.
#include <atomic>

#include "kernels.hpp"

#ifndef TTL_OPERAND_FP32
#error "strict_attention.hip belongs to the fp32 (strict) build only"
#endif

namespace {

constexpr float SCALE = 0.125f;   // head_dim^-0.5, head_dim = 64

// rows [0,T) x 64 of a q/k/v plane -> LDS [T][64]
__device__ __forceinline__ void stage_rows(float* lds, const float* g, long long ld, int T, int tid) {
    for (int q = tid; q < T * 16; q += 256) {
        const int r = q >> 4, c = q & 15;
        *(float4*)(lds + r * 64 + 4 * c) = *(const float4*)(g + (long long)r * ld + 4 * c);
    }
}
__device__ __forceinline__ void load_row(float (&v)[64], const float* g) {
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        const float4 t = *(const float4*)(g + 4 * c);
        v[4 * c] = t.x; v[4 * c + 1] = t.y; v[4 * c + 2] = t.z; v[4 * c + 3] = t.w;
    }
}
__device__ __forceinline__ float dot64(const float (&a)[64], const float* b) {
    float s = 0.f;
#pragma unroll
    for (int d = 0; d < 64; ++d) s = fmaf(a[d], b[d], s);
    return s;
}

__global__ __launch_bounds__(256) void sattn_fwd_kernel(const float* __restrict__ qkv, const QkvLayout L, float* __restrict__ out, int ldo,
                                                        float* __restrict__ lse, int T, int H, int causal) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sK = (float*)smem;
    float* sV = sK + T * 64;
    const int tid = threadIdx.x;
    const int img = blockIdx.x / H, head = blockIdx.x - img * H;
    const float* qg = qkv + (size_t)img * L.view + (size_t)head * L.head;
    stage_rows(sK, qg + L.k_off, L.tok, T, tid);
    stage_rows(sV, qg + L.v_off, L.tok, T, tid);
    __syncthreads();
    for (int q = tid; q < T; q += 256) {
        float qv[64];
        load_row(qv, qg + (long long)q * L.tok);
        const int Tk = causal ? q + 1 : T;
        float mx = -INFINITY;
        for (int j = 0; j < Tk; ++j) mx = fmaxf(mx, dot64(qv, sK + j * 64) * SCALE);
        float sum = 0.f, o[64];
#pragma unroll
        for (int d = 0; d < 64; ++d) o[d] = 0.f;
        for (int j = 0; j < Tk; ++j) {
            const float p = expf(dot64(qv, sK + j * 64) * SCALE - mx);
            sum += p;
#pragma unroll
            for (int d = 0; d < 64; ++d) o[d] = fmaf(p, sV[j * 64 + d], o[d]);
        }
        float* orow = out + (size_t)(img * T + q) * ldo + head * 64;
#pragma unroll
        for (int c = 0; c < 16; ++c) *(float4*)(orow + 4 * c) = make_float4(o[4 * c] / sum, o[4 * c + 1] / sum, o[4 * c + 2] / sum, o[4 * c + 3] / sum);
        if (lse) lse[((size_t)img * H + head) * T + q] = mx + logf(sum);
    }
}

// dQ_i = (1/8) sum_j dS_ij k_j,  dS_ij = P_ij (dO_i . v_j - dO_i . O_i),  P_ij = exp(q_i . k_j / 8 - lse_i)
__global__ __launch_bounds__(256) void sattn_bwd_dq_kernel(const float* __restrict__ qkv, const QkvLayout L, const float* __restrict__ out,
                                                           const float* __restrict__ dout, int ldo, const float* __restrict__ lse,
                                                           float* __restrict__ dqkv, int ldd, int T, int H, int causal) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sK = (float*)smem;
    float* sV = sK + T * 64;
    const int tid = threadIdx.x;
    const int img = blockIdx.x / H, head = blockIdx.x - img * H;
    const float* qg = qkv + (size_t)img * L.view + (size_t)head * L.head;
    stage_rows(sK, qg + L.k_off, L.tok, T, tid);
    stage_rows(sV, qg + L.v_off, L.tok, T, tid);
    __syncthreads();
    for (int q = tid; q < T; q += 256) {
        float qv[64], dov[64], dq[64];
        load_row(qv, qg + (long long)q * L.tok);
        load_row(dov, dout + (size_t)(img * T + q) * ldo + head * 64);
        const float delta = dot64(dov, out + (size_t)(img * T + q) * ldo + head * 64);
        const float l = lse[((size_t)img * H + head) * T + q];
#pragma unroll
        for (int d = 0; d < 64; ++d) dq[d] = 0.f;
        const int Tk = causal ? q + 1 : T;
        for (int j = 0; j < Tk; ++j) {
            const float p = expf(dot64(qv, sK + j * 64) * SCALE - l);
            const float ds = p * (dot64(dov, sV + j * 64) - delta);
#pragma unroll
            for (int d = 0; d < 64; ++d) dq[d] = fmaf(ds, sK[j * 64 + d], dq[d]);
        }
        float* o = dqkv + (size_t)(img * T + q) * ldd + head * 64;
#pragma unroll
        for (int c = 0; c < 16; ++c) *(float4*)(o + 4 * c) = make_float4(dq[4 * c] * SCALE, dq[4 * c + 1] * SCALE, dq[4 * c + 2] * SCALE, dq[4 * c + 3] * SCALE);
    }
}

// dV_j = sum_i P_ij dO_i,  dK_j = (1/8) sum_i dS_ij q_i   (two passes over the queries: one accumulator set at a time)
__global__ __launch_bounds__(256) void sattn_bwd_dkv_kernel(const float* __restrict__ qkv, const QkvLayout L, const float* __restrict__ out,
                                                            const float* __restrict__ dout, int ldo, const float* __restrict__ lse,
                                                            float* __restrict__ dqkv, int ldd, int T, int H, int need_dk, int causal) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sQ = (float*)smem;
    float* sDO = sQ + T * 64;
    float* sLse = sDO + T * 64;
    float* sDelta = sLse + T;
    const int tid = threadIdx.x;
    const int img = blockIdx.x / H, head = blockIdx.x - img * H;
    const int D = H * 64;
    const float* qg = qkv + (size_t)img * L.view + (size_t)head * L.head;
    const float* og = out + (size_t)img * T * ldo + head * 64;
    const float* dog = dout + (size_t)img * T * ldo + head * 64;
    stage_rows(sQ, qg, L.tok, T, tid);
    stage_rows(sDO, dog, ldo, T, tid);
    for (int i = tid; i < T; i += 256) {
        float d = 0.f;
        for (int e = 0; e < 64; ++e) d = fmaf(dog[(size_t)i * ldo + e], og[(size_t)i * ldo + e], d);
        sDelta[i] = d;
        sLse[i] = lse[((size_t)img * H + head) * T + i];
    }
    __syncthreads();
    for (int j = tid; j < T; j += 256) {
        float kv[64], acc[64];
        load_row(kv, qg + L.k_off + (long long)j * L.tok);
        const int i0 = causal ? j : 0;      // queries before this key never see it
        float* base = dqkv + (size_t)(img * T + j) * ldd + head * 64;
#pragma unroll
        for (int d = 0; d < 64; ++d) acc[d] = 0.f;
        for (int i = i0; i < T; ++i) {
            const float p = expf(dot64(kv, sQ + i * 64) * SCALE - sLse[i]);
#pragma unroll
            for (int d = 0; d < 64; ++d) acc[d] = fmaf(p, sDO[i * 64 + d], acc[d]);
        }
#pragma unroll
        for (int c = 0; c < 16; ++c) *(float4*)(base + 2 * D + 4 * c) = make_float4(acc[4 * c], acc[4 * c + 1], acc[4 * c + 2], acc[4 * c + 3]);
        if (!need_dk) continue;
        float vv[64];
        load_row(vv, qg + L.v_off + (long long)j * L.tok);
#pragma unroll
        for (int d = 0; d < 64; ++d) acc[d] = 0.f;
        for (int i = i0; i < T; ++i) {
            const float p = expf(dot64(kv, sQ + i * 64) * SCALE - sLse[i]);
            const float ds = p * (dot64(vv, sDO + i * 64) - sDelta[i]);
#pragma unroll
            for (int d = 0; d < 64; ++d) acc[d] = fmaf(ds, sQ[i * 64 + d], acc[d]);
        }
#pragma unroll
        for (int c = 0; c < 16; ++c)
            *(float4*)(base + D + 4 * c) = make_float4(acc[4 * c] * SCALE, acc[4 * c + 1] * SCALE, acc[4 * c + 2] * SCALE, acc[4 * c + 3] * SCALE);
    }
}

__device__ __forceinline__ float block_reduce(float v, float* red, bool is_max) {   // 256 threads; red[4]
    v = is_max ? wave_max(v) : wave_sum(v);
    __syncthreads();                 // red may still be read from the previous reduction
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return is_max ? fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])) : (red[0] + red[1]) + (red[2] + red[3]);
}

// forward for ONE query of every sequence (CLS, or qpos[sequence]): writes that row of `out` and its lse
__global__ __launch_bounds__(256) void sattn_fwd_cls_kernel(const float* __restrict__ qkv, const QkvLayout L, float* __restrict__ out, int ldo,
                                                            float* __restrict__ lse, int T, int H, const int* __restrict__ qpos, int causal) {
    __shared__ float sq[64], sp[320], red[4];
    const int tid = threadIdx.x;
    const int img = blockIdx.x / H, head = blockIdx.x - img * H;
    const float* base = qkv + (size_t)img * L.view + (size_t)head * L.head;
    const float *kbase = base + L.k_off, *vbase = base + L.v_off;
    const int qp = qpos ? qpos[img] : 0;
    const int Tk = causal ? qp + 1 : T;
    if (tid < 64) sq[tid] = base[(long long)qp * L.tok + tid];
    __syncthreads();
    float mx = -INFINITY;
    for (int j = tid; j < Tk; j += 256) {
        float s = 0.f;
        for (int d = 0; d < 64; ++d) s = fmaf(sq[d], kbase[(long long)j * L.tok + d], s);
        s *= SCALE;
        sp[j] = s;
        mx = fmaxf(mx, s);
    }
    mx = block_reduce(mx, red, true);
    float sum = 0.f;
    for (int j = tid; j < Tk; j += 256) { const float p = expf(sp[j] - mx); sp[j] = p; sum += p; }
    sum = block_reduce(sum, red, false);      // (its barriers also publish sp[])
    if (tid < 64) {
        float o = 0.f;
        for (int j = 0; j < Tk; ++j) o = fmaf(sp[j], vbase[(long long)j * L.tok + tid], o);
        out[((size_t)img * T + qp) * ldo + head * 64 + tid] = o / sum;
        if (tid == 0 && lse) lse[((size_t)img * H + head) * T + qp] = mx + logf(sum);
    }
}

// backward when d(out) is non-zero for ONE query of every sequence (attention.hip, attn_bwd_cls_kernel):
//   p_j = exp(q0.k_j/8 - lse0), dp_j = do0.v_j, ds_j = p_j (dp_j - do0.o0),
//   dq_0 = sum_j ds_j k_j / 8,  dk_j = ds_j q0 / 8,  dv_j = p_j do0;   dq_t = 0 for every other token
__global__ __launch_bounds__(256) void sattn_bwd_cls_kernel(const float* __restrict__ qkv, const QkvLayout L, const float* __restrict__ out, int ldo,
                                                            const float* __restrict__ dout_cls, const float* __restrict__ lse,
                                                            float* __restrict__ dqkv, int ldd, int T, int H, int need_dk,
                                                            const int* __restrict__ qpos, int causal) {
    __shared__ float sq[64], sdo[64], sds[320], red[4];
    const int tid = threadIdx.x;
    const int img = blockIdx.x / H, head = blockIdx.x - img * H;
    const int D = H * 64;
    const float* base = qkv + (size_t)img * L.view + (size_t)head * L.head;
    const float *kbase = base + L.k_off, *vbase = base + L.v_off;
    const int qp = qpos ? qpos[img] : 0;
    float prod = 0.f;
    if (tid < 64) {
        sq[tid] = base[(long long)qp * L.tok + tid];
        const float d = dout_cls[(size_t)img * D + head * 64 + tid];
        sdo[tid] = d;
        prod = d * out[((size_t)img * T + qp) * ldo + head * 64 + tid];
    }
    const float delta = block_reduce(prod, red, false);
    const float l0 = lse[((size_t)img * H + head) * T + qp];
    for (int j = tid; j < T; j += 256) {
        float s = 0.f, dp = 0.f;
        for (int d = 0; d < 64; ++d) {
            s = fmaf(sq[d], kbase[(long long)j * L.tok + d], s);
            dp = fmaf(sdo[d], vbase[(long long)j * L.tok + d], dp);
        }
        const bool live = !(causal && j > qp);
        const float p = live ? expf(s * SCALE - l0) : 0.f;
        const float ds = p * (dp - delta);
        sds[j] = ds;
        float* o = dqkv + (size_t)(img * T + j) * ldd + head * 64;
        for (int d = 0; d < 64; ++d) {
            o[2 * D + d] = p * sdo[d];
            if (need_dk) o[D + d] = ds * sq[d] * SCALE;
            if (j != qp) o[d] = 0.f;
        }
    }
    __syncthreads();
    if (tid < 64) {
        float v = 0.f;
        for (int j = 0; j < T; ++j) v = fmaf(sds[j], kbase[(long long)j * L.tok + tid], v);
        dqkv[((size_t)img * T + qp) * ldd + head * 64 + tid] = v * SCALE;
    }
}

}  // namespace

unsigned qkv_hm_magic(int, int) { return 0; }     // no head-major q/k/v in this build (api.hip: use_hm = 0)

hipError_t launch_attention_fwd(const op_t* qkv, QkvLayout lay, op_t* out, int ld_out, float* lse, int n, int T, int H, hipStream_t s, int causal) {
    if (T > 288) return hipErrorInvalidValue;
    const int smem = 2 * T * 64 * 4;
    static std::atomic<uint64_t> done{0};
    if (hipError_t e = ensure_smem((const void*)sattn_fwd_kernel, 2 * 288 * 64 * 4, done); e != hipSuccess) return e;
    hipLaunchKernelGGL(sattn_fwd_kernel, dim3(n * H), dim3(256), smem, s, qkv, lay, out, ld_out, lse, T, H, causal);
    return hipGetLastError();
}

hipError_t launch_attention_bwd(const op_t* qkv, QkvLayout lay, const op_t* out, const op_t* dout, int ld_o, const float* lse, op_t* dqkv,
                                int ld_dqkv, int n, int T, int H, int need_dk, hipStream_t s, int causal) {
    if (T > 288) return hipErrorInvalidValue;
    static std::atomic<uint64_t> done_q{0}, done_k{0};
    hipError_t e = ensure_smem((const void*)sattn_bwd_dq_kernel, 2 * 288 * 64 * 4, done_q);
    if (e == hipSuccess) e = ensure_smem((const void*)sattn_bwd_dkv_kernel, 2 * 288 * 64 * 4 + 2 * 288 * 4, done_k);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(sattn_bwd_dq_kernel, dim3(n * H), dim3(256), 2 * T * 64 * 4, s, qkv, lay, out, dout, ld_o, lse, dqkv, ld_dqkv, T, H, causal);
    hipLaunchKernelGGL(sattn_bwd_dkv_kernel, dim3(n * H), dim3(256), 2 * T * 64 * 4 + 2 * T * 4, s, qkv, lay, out, dout, ld_o, lse, dqkv, ld_dqkv,
                       T, H, need_dk, causal);
    return hipGetLastError();
}

hipError_t launch_attention_fwd_cls(const op_t* qkv, QkvLayout lay, op_t* out, int ld_out, float* lse, int n, int T, int H, hipStream_t s,
                                    const int* qpos, int causal) {
    if (T > 320) return hipErrorInvalidValue;
    hipLaunchKernelGGL(sattn_fwd_cls_kernel, dim3(n * H), dim3(256), 0, s, qkv, lay, out, ld_out, lse, T, H, qpos, causal);
    return hipGetLastError();
}

hipError_t launch_attention_bwd_cls(const op_t* qkv, QkvLayout lay, const op_t* out, int ld_o, const op_t* dout_cls, const float* lse,
                                    op_t* dqkv, int ld_dqkv, int n, int T, int H, int need_dk, hipStream_t s, const int* qpos, int causal) {
    if (T > 320) return hipErrorInvalidValue;
    hipLaunchKernelGGL(sattn_bwd_cls_kernel, dim3(n * H), dim3(256), 0, s, qkv, lay, out, ld_o, dout_cls, lse, dqkv, ld_dqkv, T, H, need_dk, qpos,
                       causal);
    return hipGetLastError();
}
