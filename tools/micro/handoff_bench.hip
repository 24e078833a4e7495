// What does handing a result to a second queue cost the FIRST queue?  A chain of short kernels on stream M; after each,
// stream A is told "this one is done" in one of several ways and runs a short kernel of its own.
//   hipcc --offload-arch=gfx950 -O2 tools/micro/handoff_bench.hip -o gpurun_out/handoff_bench && gpurun_out/handoff_bench
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <vector>

__global__ void spin(float *p, int iters) {
    float v = p[threadIdx.x];
    for (int i = 0; i < iters; ++i) v = v * 1.0001f + 0.5f;
    p[threadIdx.x + blockIdx.x * blockDim.x] = v;
}

// first queue: the LAST thing the kernel does is publish its index; second queue: must see at least its own
__global__ void spin_publish(float *p, int iters, int *seq, int idx) {
    float v = p[threadIdx.x];
    for (int i = 0; i < iters; ++i) v = v * 1.0001f + 0.5f;
    p[threadIdx.x + blockIdx.x * blockDim.x] = v;
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) *seq = idx;
}
__global__ void spin_check(float *p, int iters, const int *seq, int idx, int *violations) {
    if (blockIdx.x == 0 && threadIdx.x == 0 && *seq < idx) atomicAdd(violations, 1);
    float v = p[threadIdx.x];
    for (int i = 0; i < iters; ++i) v = v * 1.0001f + 0.5f;
    p[threadIdx.x + blockIdx.x * blockDim.x] = v;
}

#define CK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(err_)); return 1; } } while (0)

int main() {
    const int N = 400, ITERS = 600, WGS = 512;
    float *buf, *buf2;
    CK(hipMalloc(&buf, WGS * 256 * 4 * 2));
    buf2 = buf + WGS * 256;
    uint32_t *flag;
    CK(hipMalloc(&flag, 4 * (N + 1)));
    CK(hipMemset(flag, 0, 4 * (N + 1)));
    int *seq;
    CK(hipMalloc(&seq, 8));
    hipStream_t M, A;
    CK(hipStreamCreate(&M));
    CK(hipStreamCreate(&A));
    std::vector<hipEvent_t> ev(N);
    for (auto &e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    hipEvent_t join;
    CK(hipEventCreateWithFlags(&join, hipEventDisableTiming));
    const char *names[] = {"plain chain", "record after each kernel (nobody waits)", "record + second queue waits and runs a kernel",
                           "stop event on the launch itself (hipExtLaunchKernelGGL) + second queue", "record every 4th kernel + second queue",
                           "hipStreamWriteValue32 / hipStreamWaitValue32 + second queue", "no hand-off: the second queue's kernels run unordered"};
    for (int rep = 0; rep < 2; ++rep)
        for (int mode = 0; mode < 7; ++mode) {
            CK(hipDeviceSynchronize());
            CK(hipMemset(flag, 0, 4 * (N + 1)));
            CK(hipMemset(seq, 0, 8));
            CK(hipDeviceSynchronize());
            auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < N; ++i) {
                if (mode == 3) {
                    hipExtLaunchKernelGGL(spin_publish, dim3(WGS), dim3(256), 0, M, nullptr, ev[i], 0, buf, ITERS, seq, i + 1);
                } else {
                    hipLaunchKernelGGL(spin_publish, dim3(WGS), dim3(256), 0, M, buf, ITERS, seq, i + 1);
                }
                const bool hand = mode == 4 ? (i % 4 == 3) : true;
                if ((mode == 1 || mode == 2 || mode == 4) && hand) CK(hipEventRecord(ev[i], M));
                if ((mode == 2 || mode == 3 || mode == 4) && hand) {
                    CK(hipStreamWaitEvent(A, ev[i], 0));
                    hipLaunchKernelGGL(spin_check, dim3(64), dim3(256), 0, A, buf2, ITERS / 2, seq, i + 1, seq + 1);
                }
                if (mode == 6) hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, A, buf2, ITERS / 2);
                if (mode == 5) {
                    CK(hipStreamWriteValue32(M, flag + i, 1, 0));
                    CK(hipStreamWaitValue32(A, flag + i, 1, hipStreamWaitValueGte, 0xffffffffu));
                    hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, A, buf2, ITERS / 2);
                }
            }
            CK(hipStreamSynchronize(M));
            auto t1 = std::chrono::steady_clock::now();
            CK(hipEventRecord(join, A));
            CK(hipStreamWaitEvent(M, join, 0));
            CK(hipDeviceSynchronize());
            int viol[2];
            CK(hipMemcpy(viol, seq, 8, hipMemcpyDeviceToHost));
            if (rep) printf("%-82s %7.2f us per kernel of the first queue, %d ordering violations\n", names[mode], std::chrono::duration<double, std::micro>(t1 - t0).count() / N, viol[1]);
        }
    return 0;
}
