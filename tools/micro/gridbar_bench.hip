// Is a grid-wide barrier inside one kernel cheaper than a kernel boundary?  Phase A writes a tensor and a per-workgroup
// partial; phase B needs ALL partials (a BatchNorm's statistics) and rewrites the workgroup's own part of the tensor.
//   two launches: A | B           one launch: A, ticket barrier (bounded spin), B
//   hipcc --offload-arch=gfx950 -O2 tools/micro/gridbar_bench.hip -o /tmp/gb && /tmp/gb
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>

#define CK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(err_)); return 1; } } while (0)

constexpr int kPer = 8;   // float4 per thread

__device__ __forceinline__ float phase_a(float4 *t, float *part, int iters, float4 (&keep)[kPer]) {
    const int tid = threadIdx.x, wg = blockIdx.x;
    float s = 0.f;
    for (int q = 0; q < kPer; ++q) {
        float4 v = t[((long)wg * kPer + q) * 256 + tid];
        for (int i = 0; i < iters; ++i) v.x = v.x * 1.0001f + 0.5f;
        v.y += 1.f;
        keep[q] = v;
        t[((long)wg * kPer + q) * 256 + tid] = v;
        s += v.y;
    }
    __shared__ float red[256];
    red[tid] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) red[tid] += red[tid + o];
        __syncthreads();
    }
    if (tid == 0) part[wg] = red[0];
    return red[0];
}

__device__ __forceinline__ float all_parts(const float *part, int nwg) {
    __shared__ float red2[256];
    float s = 0.f;
    for (int i = threadIdx.x; i < nwg; i += 256) s += __builtin_nontemporal_load(part + i);
    red2[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red2[threadIdx.x] += red2[threadIdx.x + o];
        __syncthreads();
    }
    return red2[0];
}

__global__ __launch_bounds__(256) void k_a(float4 *t, float *part, int iters) {
    float4 keep[kPer];
    phase_a(t, part, iters, keep);
}
__global__ __launch_bounds__(256) void k_b(float4 *t, float4 *out, const float *part, int nwg) {
    const float m = all_parts(part, nwg) / (float)(nwg * 256 * kPer);
    for (int q = 0; q < kPer; ++q) {
        float4 v = t[((long)blockIdx.x * kPer + q) * 256 + threadIdx.x];
        v.y -= m;
        out[((long)blockIdx.x * kPer + q) * 256 + threadIdx.x] = v;
    }
}
__global__ __launch_bounds__(256) void k_ab(float4 *t, float4 *out, float *part, int iters, unsigned *counter, int *timeouts) {
    float4 keep[kPer];
    phase_a(t, part, iters, keep);
    // ---- grid barrier: every workgroup of the launch is resident (grid <= slots); bounded wait all the same
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        const unsigned ticket = atomicAdd(counter, 1u);
        const unsigned target = (ticket / gridDim.x + 1u) * gridDim.x;
        int spins = 0;
        while ((int)(__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > 2000000) {
                atomicAdd(timeouts, 1);
                break;
            }
        }
        __threadfence();
    }
    __syncthreads();
    const float m = all_parts(part, gridDim.x) / (float)(gridDim.x * 256 * kPer);
    for (int q = 0; q < kPer; ++q) {
        float4 v = keep[q];
        v.y -= m;
        out[((long)blockIdx.x * kPer + q) * 256 + threadIdx.x] = v;
    }
}

int main() {
    const int N = 300;
    for (int nwg : {256, 512}) {
        const size_t n4 = (size_t)nwg * kPer * 256;
        float4 *t, *out;
        float *part;
        unsigned *counter;
        int *timeouts;
        CK(hipMalloc(&t, n4 * 16));
        CK(hipMalloc(&out, n4 * 16));
        CK(hipMalloc(&part, nwg * 4));
        CK(hipMalloc(&counter, 4));
        CK(hipMalloc(&timeouts, 4));
        CK(hipMemset(t, 0, n4 * 16));
        CK(hipMemset(counter, 0, 4));
        CK(hipMemset(timeouts, 0, 4));
        hipStream_t s;
        CK(hipStreamCreate(&s));
        for (int iters : {0, 2000}) {
            double us[2];
            for (int mode = 0; mode < 2; ++mode) {
                for (int rep = 0; rep < 2; ++rep) {
                    CK(hipStreamSynchronize(s));
                    auto t0 = std::chrono::steady_clock::now();
                    for (int i = 0; i < N; ++i) {
                        if (mode == 0) {
                            hipLaunchKernelGGL(k_a, dim3(nwg), dim3(256), 0, s, t, part, iters);
                            hipLaunchKernelGGL(k_b, dim3(nwg), dim3(256), 0, s, t, out, part, nwg);
                        } else {
                            hipLaunchKernelGGL(k_ab, dim3(nwg), dim3(256), 0, s, t, out, part, iters, counter, timeouts);
                        }
                    }
                    CK(hipStreamSynchronize(s));
                    us[mode] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
                }
            }
            int to = 0;
            CK(hipMemcpy(&to, timeouts, 4, hipMemcpyDeviceToHost));
            // check: out.y == t.y - mean(t.y) with t.y uniform = launches so far -> 0
            float4 h;
            CK(hipMemcpy(&h, out + 5, 16, hipMemcpyDeviceToHost));
            printf("%d workgroups, %.1f MB tensor, phase A spin %d: two launches %.2f us, one launch with a grid barrier %.2f us (timeouts %d, out.y %.3f)\n",
                   nwg, n4 * 16 / 1e6, iters, us[0], us[1], to, h.y);
        }
    }
    return 0;
}
