// Experiment (not part of libatx): how the launch shape of a plain streaming y = a*x + b kernel changes the rate on MI355X.
//   shape 0: one 16-B vector per thread, no loop, grid = n_vec / 256            (what torch's elementwise kernels do)
//   shape 1: U vectors per thread (U independent loads), no loop, grid = n_vec / (256 U)
//   shape 2: grid-stride loop over chunks of 256 U vectors, grid capped
#include <hip/hip_runtime.h>
#include <stdint.h>

struct alignas(16) V4 { float v[4]; };

template <int U>
__global__ void __launch_bounds__(256) k_noloop(const V4* __restrict__ x, V4* __restrict__ y, int64_t n, float a, float b) {
    const int64_t base = (int64_t)blockIdx.x * 256 * U + threadIdx.x;
    V4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) if (base + u * 256 < n) v[u] = x[base + u * 256];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        if (base + u * 256 >= n) continue;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[u].v[e] = v[u].v[e] * a + b;
        y[base + u * 256] = v[u];
    }
}

template <int U>
__global__ void __launch_bounds__(256) k_loop(const V4* __restrict__ x, V4* __restrict__ y, int64_t n, float a, float b) {
    for (int64_t base = (int64_t)blockIdx.x * 256 * U + threadIdx.x; base < n; base += (int64_t)gridDim.x * 256 * U) {
        V4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) if (base + u * 256 < n) v[u] = x[base + u * 256];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (base + u * 256 >= n) continue;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[u].v[e] = v[u].v[e] * a + b;
            y[base + u * 256] = v[u];
        }
    }
}

struct Op { int op; int use_mask; double p0; double p1; };

// shape 3: shape 0 plus what a per-level program needs per vector: column = vi % C, one 24-byte operator per stage from a
// small global table, a switch on the operator, optional point mask byte
__global__ void __launch_bounds__(256) k_table(const V4* __restrict__ x, V4* __restrict__ y, int64_t n, int C, const Op* __restrict__ table,
                                               int n_stage, const uint8_t* __restrict__ mask) {
    const int64_t vi = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (vi >= n) return;
    const unsigned row = (unsigned)(vi / C);
    const int c = (int)(vi - (int64_t)row * C);
    V4 v = x[vi];
    const bool masked = mask ? mask[row] != 0 : false;
    for (int s = 0; s < n_stage; ++s) {
        const Op o = table[s * C + c];
        const float p0 = (float)o.p0, p1 = (float)o.p1;
        switch (o.op) {
            case 1:
#pragma unroll
                for (int e = 0; e < 4; ++e) v.v[e] = v.v[e] * p0 + p1;
                break;
            case 2:
#pragma unroll
                for (int e = 0; e < 4; ++e) v.v[e] = (v.v[e] - p1) / p0;
                break;
            case 3:
#pragma unroll
                for (int e = 0; e < 4; ++e) v.v[e] = v.v[e] * p0;
                break;
            default: break;
        }
        if (o.use_mask && masked) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v.v[e] = __uint_as_float(0x7fc00000u);
        }
    }
    y[vi] = v;
}

struct Op12 { int op_mask; float p0; float p1; };

// shape 4: shape 3 with 12-byte operators (op | use_mask << 16, p0, p1 as float), all stages fetched before the data
__global__ void __launch_bounds__(256) k_table12(const V4* __restrict__ x, V4* __restrict__ y, int64_t n, int C, const Op12* __restrict__ table,
                                                 int n_stage, const uint8_t* __restrict__ mask) {
    const int64_t vi = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (vi >= n) return;
    const unsigned row = (unsigned)(vi / C);
    const int c = (int)(vi - (int64_t)row * C);
    Op12 ops[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) if (s < n_stage) ops[s] = table[s * C + c];
    V4 v = x[vi];
    const bool masked = mask ? mask[row] != 0 : false;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        if (s >= n_stage) break;
        const float p0 = ops[s].p0, p1 = ops[s].p1;
        switch (ops[s].op_mask & 0xffff) {
            case 1:
#pragma unroll
                for (int e = 0; e < 4; ++e) v.v[e] = v.v[e] * p0 + p1;
                break;
            case 2:
#pragma unroll
                for (int e = 0; e < 4; ++e) v.v[e] = (v.v[e] - p1) / p0;
                break;
            case 3:
#pragma unroll
                for (int e = 0; e < 4; ++e) v.v[e] = v.v[e] * p0;
                break;
            default: break;
        }
        if ((ops[s].op_mask >> 16) && masked) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v.v[e] = __uint_as_float(0x7fc00000u);
        }
    }
    y[vi] = v;
}

extern "C" int run_table12(const void* x, void* y, int64_t n_vec, int C, const void* table, int n_stage, const void* mask, void* stream) {
    const int64_t blocks = (n_vec + 255) / 256;
    hipLaunchKernelGGL(k_table12, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const V4*)x, (V4*)y, n_vec, C, (const Op12*)table, n_stage,
                       (const uint8_t*)mask);
    return (int)hipGetLastError();
}

extern "C" int run_table(const void* x, void* y, int64_t n_vec, int C, const void* table, int n_stage, const void* mask, void* stream) {
    const int64_t blocks = (n_vec + 255) / 256;
    hipLaunchKernelGGL(k_table, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const V4*)x, (V4*)y, n_vec, C, (const Op*)table, n_stage,
                       (const uint8_t*)mask);
    return (int)hipGetLastError();
}

extern "C" int run_shape(int shape, int U, int64_t grid_cap, const void* x, void* y, int64_t n_vec, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    const V4* xv = (const V4*)x;
    V4* yv = (V4*)y;
    int64_t blocks = (n_vec + 256 * U - 1) / (256 * U);
    if (shape == 2 && blocks > grid_cap) blocks = grid_cap;
#define LAUNCH(K, UU) hipLaunchKernelGGL((K<UU>), dim3((unsigned)blocks), dim3(256), 0, s, xv, yv, n_vec, 2.0f, 1.0f)
    if (shape != 2) {
        if (U == 1) LAUNCH(k_noloop, 1); else if (U == 2) LAUNCH(k_noloop, 2); else if (U == 4) LAUNCH(k_noloop, 4); else LAUNCH(k_noloop, 8);
    } else {
        if (U == 1) LAUNCH(k_loop, 1); else if (U == 2) LAUNCH(k_loop, 2); else if (U == 4) LAUNCH(k_loop, 4); else LAUNCH(k_loop, 8);
    }
    return (int)hipGetLastError();
}
