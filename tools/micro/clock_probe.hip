// Engine-clock probe: one wave spins for `spin` ticks of the constant 100-MHz counter (s_memrealtime) and reports how
// many shader-clock cycles (s_memtime) passed meanwhile -> the clock the chip held during that window.  Launched back to
// back on a stream of its own beside the step (tools/clock_in_step.py): is a step slower than the sum of its kernels
// measured alone because the chip lowers its clock when several queues keep every CU busy?
//   hipcc --offload-arch=gfx950 -O2 -shared -fPIC tools/micro/clock_probe.hip -o tools/micro/libclockprobe.so
#include <hip/hip_runtime.h>

__global__ void clock_probe_kernel(unsigned long long *out, int idx, int spin) {
    if (threadIdx.x != 0) return;
    const unsigned long long w0 = wall_clock64(), c0 = clock64();
    unsigned long long w = w0;
    while ((long long)(w - w0) < spin) {
        __builtin_amdgcn_s_sleep(8);
        w = wall_clock64();
    }
    const unsigned long long c1 = clock64();
    out[idx * 3 + 0] = w0;
    out[idx * 3 + 1] = w - w0;
    out[idx * 3 + 2] = c1 - c0;
}

extern "C" int clock_probe_launch(unsigned long long *out, int idx, int spin, void *stream) {
    hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), out, idx, spin);
    return (int)hipGetLastError();
}

__global__ void wall_now_kernel(unsigned long long *out) { out[0] = wall_clock64(); }
extern "C" int clock_probe_now(unsigned long long *out, void *stream) {
    hipLaunchKernelGGL(wall_now_kernel, dim3(1), dim3(1), 0, static_cast<hipStream_t>(stream), out);
    return (int)hipGetLastError();
}
