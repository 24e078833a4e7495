// Shared device helpers for the combat_hip kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

#include "combat_hip.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
// native 16-byte register type for staging arrays: copying HIP's uint4 *struct* from global memory
// into a local array lowers to an address-space-crossing memcpy that SROA does not promote, which
// leaves the array in scratch memory (and every prefetch waits for its own load)
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;

// Every kernel launch of the library goes through COMBAT_LAUNCH.  Normally it is hipLaunchKernelGGL.  While the plan
// replayer (plan.cpp) has set this thread's `t_stop`, the launch carries that event as its completion event
// (hipExtLaunchKernelGGL's stopEvent): the kernel's own dispatch packet signals it.  That is how a result is handed to
// the auxiliary queue: an hipEventRecord between two kernels is a packet of its own in the producing queue and holds
// the NEXT kernel of that queue back by 3-5 us (tools/micro/handoff_bench.hip: 11.8 us per kernel plain, 17.0 with
// record + wait, 12.4 with the event on the launch) -- ~34 hand-offs sit on the critical queue of an alternated step.
namespace combat_launch {
extern thread_local hipEvent_t t_stop;
extern thread_local bool t_stop_used;
}  // namespace combat_launch
#define COMBAT_LAUNCH(kernel, grid, block, smem, stream, ...)                                             \
    do {                                                                                                  \
        if (hipEvent_t stop_ = combat_launch::t_stop) {                                                   \
            hipExtLaunchKernelGGL(kernel, grid, block, smem, stream, nullptr, stop_, 0, __VA_ARGS__);     \
            combat_launch::t_stop_used = true;                                                            \
        } else {                                                                                          \
            hipLaunchKernelGGL(kernel, grid, block, smem, stream, __VA_ARGS__);                           \
        }                                                                                                 \
    } while (0)

#define CB_LAUNCH_CHECK()                                  \
    do {                                                   \
        if (hipGetLastError() != hipSuccess) return COMBAT_ELAUNCH; \
    } while (0)

__device__ __forceinline__ float bf16_bits_to_f32(uint32_t b) { return __uint_as_float(b << 16); }

// fp32 -> bf16 (RNE; NaN stays NaN): a plain cast lowers to v_cvt_pk_bf16_f32 on gfx950
__device__ __forceinline__ uint16_t f32_to_bf16_bits(float f) {
    __bf16 h = (__bf16)f;
    return __builtin_bit_cast(uint16_t, h);
}

__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    return (uint32_t)f32_to_bf16_bits(lo) | ((uint32_t)f32_to_bf16_bits(hi) << 16);
}

__device__ __forceinline__ void unpack8(const uint4 &u, float (&v)[8]) {
    v[0] = bf16_bits_to_f32(u.x & 0xffffu);
    v[1] = bf16_bits_to_f32(u.x >> 16);
    v[2] = bf16_bits_to_f32(u.y & 0xffffu);
    v[3] = bf16_bits_to_f32(u.y >> 16);
    v[4] = bf16_bits_to_f32(u.z & 0xffffu);
    v[5] = bf16_bits_to_f32(u.z >> 16);
    v[6] = bf16_bits_to_f32(u.w & 0xffffu);
    v[7] = bf16_bits_to_f32(u.w >> 16);
}

__device__ __forceinline__ void unpack8v(const u32x4_t &u, float (&v)[8]) {
    v[0] = bf16_bits_to_f32(u[0] & 0xffffu);
    v[1] = bf16_bits_to_f32(u[0] >> 16);
    v[2] = bf16_bits_to_f32(u[1] & 0xffffu);
    v[3] = bf16_bits_to_f32(u[1] >> 16);
    v[4] = bf16_bits_to_f32(u[2] & 0xffffu);
    v[5] = bf16_bits_to_f32(u[2] >> 16);
    v[6] = bf16_bits_to_f32(u[3] & 0xffffu);
    v[7] = bf16_bits_to_f32(u[3] >> 16);
}

__device__ __forceinline__ u32x4_t pack8v(const float (&v)[8]) {
    u32x4_t u;
    u[0] = pack_bf16x2(v[0], v[1]);
    u[1] = pack_bf16x2(v[2], v[3]);
    u[2] = pack_bf16x2(v[4], v[5]);
    u[3] = pack_bf16x2(v[6], v[7]);
    return u;
}

__device__ __forceinline__ uint4 pack8(const float (&v)[8]) {
    uint4 u;
    u.x = pack_bf16x2(v[0], v[1]);
    u.y = pack_bf16x2(v[2], v[3]);
    u.z = pack_bf16x2(v[4], v[5]);
    u.w = pack_bf16x2(v[6], v[7]);
    return u;
}

__device__ __forceinline__ void load8f(const float *p, float (&v)[8]) {
    const float4 a = *reinterpret_cast<const float4 *>(p);
    const float4 b = *reinterpret_cast<const float4 *>(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
    v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}

__device__ __forceinline__ float round_bf16(float f) { return bf16_bits_to_f32(f32_to_bf16_bits(f)); }

static inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

// combat_set_deterministic / COMBAT_DETERMINISTIC=1 (capi.cpp): launches pick their order-independent forms
bool combat_deterministic();
// <= 2 MB of library-owned scratch per stream for two-stage reductions (capi.cpp); nullptr if `bytes` exceeds it
float *combat_stream_scratch(void *stream, size_t bytes);

static inline int ilog2_exact(int v) {
    if (v <= 0 || (v & (v - 1))) return -1;
    int s = 0;
    while ((1 << s) < v) ++s;
    return s;
}
