// RCCL behind the C ABI (combat_allreduce(buf, count, dtype, comm, stream)), for a host that is not
// PyTorch: the repo's own data-parallel path exchanges gradients through torch.distributed (backend "nccl" = RCCL,
// combat_amd/dist.py), which north_star allows; these entry points give a C / C++ / other-FFI host the same collective
// on the same flat gradient buffers.
//
// RCCL is resolved at FIRST USE with dlsym -- first among the libraries the process has already loaded (a PyTorch
// process carries its own librccl.so; a second copy must not be pulled in beside it), else dlopen("librccl.so") -- so
// libcombat_hip.so has no link-time dependency on it and loads on a box without RCCL.
//
// Replaces nothing in the reference (it has no distributed code: train_generator.py imports none); the call sites are the
// two gradient exchanges of the data-parallel step added around train_generator.py:208-212 and :253-255.
#include <dlfcn.h>
#include <stddef.h>
#include <string.h>

#include "combat_hip.h"

namespace {

struct CombatUid {
    char internal[COMBAT_COMM_UNIQUE_ID_BYTES];      // = ncclUniqueId (NCCL_UNIQUE_ID_BYTES), passed by value
};

typedef int (*get_unique_id_fn)(void *);
typedef int (*comm_init_rank_fn)(void **, int, CombatUid, int);
typedef int (*comm_destroy_fn)(void *);
typedef int (*all_reduce_fn)(const void *, void *, size_t, int, int, void *, void *);

struct Rccl {
    get_unique_id_fn get_unique_id = nullptr;
    comm_init_rank_fn comm_init_rank = nullptr;
    comm_destroy_fn comm_destroy = nullptr;
    all_reduce_fn all_reduce = nullptr;
    bool ok = false;
};

const Rccl &rccl() {
    static const Rccl r = [] {
        Rccl x;
        void *h = RTLD_DEFAULT;
        if (!dlsym(h, "ncclAllReduce")) {
            h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
            if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
            if (!h) return x;
        }
        x.get_unique_id = reinterpret_cast<get_unique_id_fn>(dlsym(h, "ncclGetUniqueId"));
        x.comm_init_rank = reinterpret_cast<comm_init_rank_fn>(dlsym(h, "ncclCommInitRank"));
        x.comm_destroy = reinterpret_cast<comm_destroy_fn>(dlsym(h, "ncclCommDestroy"));
        x.all_reduce = reinterpret_cast<all_reduce_fn>(dlsym(h, "ncclAllReduce"));
        x.ok = x.get_unique_id && x.comm_init_rank && x.comm_destroy && x.all_reduce;
        return x;
    }();
    return r;
}

}  // namespace

extern "C" int combat_comm_unique_id(void *out128) {
    if (!out128) return COMBAT_EINVAL;
    const Rccl &r = rccl();
    if (!r.ok) return COMBAT_ELAUNCH;
    return r.get_unique_id(out128) == 0 ? COMBAT_OK : COMBAT_ELAUNCH;
}

extern "C" int combat_comm_init_rank(void **comm, int32_t nranks, const void *unique_id128, int32_t rank) {
    if (!comm || !unique_id128 || nranks <= 0 || rank < 0 || rank >= nranks) return COMBAT_EINVAL;
    const Rccl &r = rccl();
    if (!r.ok) return COMBAT_ELAUNCH;
    CombatUid uid;
    memcpy(uid.internal, unique_id128, sizeof(uid.internal));
    return r.comm_init_rank(comm, nranks, uid, rank) == 0 ? COMBAT_OK : COMBAT_ELAUNCH;
}

extern "C" int combat_comm_destroy(void *comm) {
    if (!comm) return COMBAT_EINVAL;
    const Rccl &r = rccl();
    if (!r.ok) return COMBAT_ELAUNCH;
    return r.comm_destroy(comm) == 0 ? COMBAT_OK : COMBAT_ELAUNCH;
}

// in-place sum over the ranks of `comm`; dtype: COMBAT_DTYPE_F32 (the flat gradient buffers) or COMBAT_DTYPE_BF16
extern "C" int combat_allreduce(void *buf, int64_t count, int32_t dtype, void *comm, void *stream) {
    if (!buf || count <= 0 || !comm || (dtype != COMBAT_DTYPE_F32 && dtype != COMBAT_DTYPE_BF16)) return COMBAT_EINVAL;
    const Rccl &r = rccl();
    if (!r.ok) return COMBAT_ELAUNCH;
    const int nccl_type = dtype == COMBAT_DTYPE_F32 ? 7 /* ncclFloat32 */ : 9 /* ncclBfloat16 */;
    return r.all_reduce(buf, buf, (size_t)count, nccl_type, 0 /* ncclSum */, comm, stream) == 0 ? COMBAT_OK : COMBAT_ELAUNCH;
}
