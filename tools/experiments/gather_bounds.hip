// Experiment (not part of libatx): where the time of the headline gather goes.  The direct kernel's loop body in three forms:
//   mode 0: the real thing (k = 4 weighted gather, f32, 16-byte vectors, column stacks)
//   mode 1: loads only — the result is stored only if it equals an impossible bit pattern, so the loads cannot be elided
//   mode 2: stores only — no index or source reads, a constant is stored
#include <hip/hip_runtime.h>
#include <stdint.h>

struct alignas(16) V4 { float v[4]; };

template <int MODE>
__global__ void __launch_bounds__(256) k_gather(const float* __restrict__ src, float* __restrict__ out, const int32_t* __restrict__ idx,
                                                const float* __restrict__ w, int64_t n_items, int C, int64_t src_pitch, int64_t out_pitch) {
    const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (q >= n_items) return;
    const unsigned t = (unsigned)((uint64_t)q / (unsigned)C);
    const int c = (int)(q - (int64_t)t * C);
    V4 acc;
    for (int e = 0; e < 4; ++e) acc.v[e] = 0.0f;
    if (MODE != 2) {
        int32_t p[4];
        float wv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            p[j] = __builtin_nontemporal_load(idx + (int64_t)t * 4 + j);
            wv[j] = __builtin_nontemporal_load(w + (int64_t)t * 4 + j);
        }
        V4 v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = *reinterpret_cast<const V4*>(src + (int64_t)p[j] * src_pitch + (int64_t)c * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc.v[e] = acc.v[e] + wv[j] * v[j].v[e];
    }
    if (MODE == 1) {
        if (__float_as_uint(acc.v[0]) != 0x7fc12345u) return;  // never true for real data
    }
    typedef float NV __attribute__((ext_vector_type(4)));
    __builtin_nontemporal_store(*reinterpret_cast<const NV*>(&acc), reinterpret_cast<NV*>(out + (int64_t)t * out_pitch + (int64_t)c * 4));
}

extern "C" int run_gather(int mode, const void* src, void* out, const void* idx, const void* w, int64_t n_tgt, int C, int64_t src_pitch,
                          int64_t out_pitch, void* stream) {
    const int64_t n_items = n_tgt * C;
    const unsigned blocks = (unsigned)((n_items + 255) / 256);
    hipStream_t s = (hipStream_t)stream;
#define GO(M) hipLaunchKernelGGL(k_gather<M>, dim3(blocks), dim3(256), 0, s, (const float*)src, (float*)out, (const int32_t*)idx, (const float*)w, \
                                 n_items, C, src_pitch, out_pitch)
    if (mode == 0) GO(0); else if (mode == 1) GO(1); else GO(2);
    return (int)hipGetLastError();
}
