// C-side replay of a recorded call sequence ("plan"): the launch list of one network pass is walked here, not in
// Python -- one foreign call per plan instead of one per kernel, hand-off events created once and reused.
//
// The host mirror (combat_amd/engine.py::Plan) builds a plan by arming combat_plan_record(plan, queue) and then making
// the ordinary C-ABI call with a null stream: the entry point's COMBAT_PLAN_HOOK captures it.  queue < 0: the call
// runs on the plan's own stream; queue >= 0: it may run beside the calls that follow it (weight gradients), on
// auxiliary stream `queue`, ordered after everything recorded before it by an event.  combat_plan_run replays a
// range of calls; combat_plan_join makes the plan's own stream wait for the auxiliary streams it used (at a
// data-parallel all-reduce mark, and at the end of the plan).
#include <hip/hip_runtime.h>

#include <stdlib.h>

#include <vector>

#include "combat_hip.h"
#include "common.hpp"
#include "plan.hpp"

struct combat_plan {
    struct Call {
        std::function<int(void *)> fn;
        int queue;
        int after;        // index of an earlier call this one waits for (-1: none)
        bool awaited;     // some later call waits for this one: record `done` behind it
    };
    std::vector<Call> calls;
    std::vector<hipEvent_t> handoff;   // one per call that runs on an auxiliary queue
    std::vector<hipEvent_t> done;      // one per awaited call
    std::vector<hipEvent_t> join;      // one per auxiliary queue
    unsigned used = 0;                 // auxiliary queues with work not yet joined
    int failed = -1;                   // index of the call whose status combat_plan_run returned
};

namespace combat_launch {
thread_local hipEvent_t t_stop = nullptr;
thread_local bool t_stop_used = false;
}  // namespace combat_launch

namespace {
thread_local combat_plan *t_plan = nullptr;
thread_local int t_queue = -1;

hipEvent_t new_event() {
    hipEvent_t e = nullptr;
    return hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess ? e : nullptr;
}
}  // namespace

namespace combat_plan_detail {
bool armed() { return t_plan != nullptr; }
int capture(std::function<int(void *)> call) {
    combat_plan *p = t_plan;
    t_plan = nullptr;   // one-shot: a captured call that itself calls entry points replays them normally
    p->calls.push_back({std::move(call), t_queue, -1, false});
    p->handoff.push_back(nullptr);
    p->done.push_back(nullptr);
    return COMBAT_OK;
}
}  // namespace combat_plan_detail

extern "C" combat_plan *combat_plan_create(void) { return new (std::nothrow) combat_plan(); }

extern "C" void combat_plan_destroy(combat_plan *p) {
    if (!p) return;
    for (hipEvent_t e : p->handoff)
        if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : p->join)
        if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : p->done)
        if (e) (void)hipEventDestroy(e);
    delete p;
}

extern "C" int combat_plan_record(combat_plan *p, int32_t queue) {
    if (!p || t_plan) return COMBAT_EINVAL;
    t_plan = p;
    t_queue = queue;
    return COMBAT_OK;
}

extern "C" int combat_plan_set_after(combat_plan *p, int32_t call_index) {
    if (!p || p->calls.empty() || call_index < 0 || call_index >= (int32_t)p->calls.size() - 1) return COMBAT_EINVAL;
    p->calls.back().after = call_index;
    p->calls[call_index].awaited = true;
    return COMBAT_OK;
}

extern "C" int combat_plan_record_cancel(void) {   // the armed call was not a hooked entry point
    const bool was = t_plan != nullptr;
    t_plan = nullptr;
    return was ? 1 : 0;
}

extern "C" int32_t combat_plan_size(const combat_plan *p) { return p ? (int32_t)p->calls.size() : COMBAT_EINVAL; }

extern "C" int32_t combat_plan_failed_call(const combat_plan *p) { return p ? p->failed : COMBAT_EINVAL; }

extern "C" int combat_plan_run(combat_plan *p, int32_t begin, int32_t end, void *stream, void *const *aux_streams, int32_t n_aux) {
    if (!p || begin < 0 || end > (int32_t)p->calls.size() || begin > end || n_aux < 0 || (n_aux && !aux_streams)) return COMBAT_EINVAL;
    hipStream_t main_st = reinterpret_cast<hipStream_t>(stream);
    // Hand-off to an auxiliary queue: the aux call must see everything enqueued before it.  The cheap form: the plan-stream
    // call right in front of it is launched with the hand-off event as its kernel's completion event (COMBAT_LAUNCH,
    // common.hpp) -- nothing extra enters the plan's own queue.  Fallback (the call in front launched no kernel, or the
    // aux call opens the range): an event record on the plan's stream.
    static const bool record_always = getenv("COMBAT_HANDOFF_RECORD") && atoi(getenv("COMBAT_HANDOFF_RECORD")) == 1;
    hipEvent_t carried = nullptr;     // completion event of the last plan-stream call, if it was launched with one
    for (int32_t i = begin; i < end; ++i) {
        const combat_plan::Call &c = p->calls[i];
        const bool on_aux = c.queue >= 0 && n_aux > 0;
        void *const st_i = on_aux ? aux_streams[c.queue % n_aux] : stream;
        // (in-line replay, n_aux == 0: stream order already gives every `after`)
        if (c.after >= 0 && n_aux > 0 && p->done[c.after] &&
            hipStreamWaitEvent(reinterpret_cast<hipStream_t>(st_i), p->done[c.after], 0) != hipSuccess)
            return COMBAT_ELAUNCH;
        int rc;
        if (on_aux) {
            hipStream_t aux = reinterpret_cast<hipStream_t>(st_i);
            hipEvent_t ev = carried;
            if (!ev) {
                hipEvent_t &own = p->handoff[i];
                if (!own && !(own = new_event())) return COMBAT_ELAUNCH;
                if (hipEventRecord(own, main_st) != hipSuccess) return COMBAT_ELAUNCH;
                ev = own;
            }
            // everything enqueued so far (the producers of this call's operands) happens-before it
            if (hipStreamWaitEvent(aux, ev, 0) != hipSuccess) return COMBAT_ELAUNCH;
            rc = c.fn(st_i);
            p->used |= 1u << (c.queue % n_aux);
        } else {
            const bool next_aux = !record_always && n_aux > 0 && i + 1 < end && p->calls[i + 1].queue >= 0;
            hipEvent_t ev = nullptr;
            if (next_aux) {
                hipEvent_t &own = p->handoff[i + 1];
                if (!own && !(own = new_event())) return COMBAT_ELAUNCH;
                ev = own;
                combat_launch::t_stop = ev;
                combat_launch::t_stop_used = false;
            }
            rc = c.fn(stream);
            combat_launch::t_stop = nullptr;
            carried = next_aux && combat_launch::t_stop_used ? ev : nullptr;
        }
        if (rc == COMBAT_OK && c.awaited && n_aux > 0) {
            hipEvent_t &ev = p->done[i];
            if (!ev && !(ev = new_event())) return COMBAT_ELAUNCH;
            if (hipEventRecord(ev, reinterpret_cast<hipStream_t>(st_i)) != hipSuccess) return COMBAT_ELAUNCH;
        }
        if (rc != COMBAT_OK) {
            p->failed = i;
            return rc;
        }
    }
    return COMBAT_OK;
}

extern "C" int combat_plan_join(combat_plan *p, void *stream, void *const *aux_streams, int32_t n_aux) {
    if (!p || n_aux < 0 || n_aux > 32 || (n_aux && !aux_streams)) return COMBAT_EINVAL;
    hipStream_t main_st = reinterpret_cast<hipStream_t>(stream);
    if ((int32_t)p->join.size() < n_aux) p->join.resize(n_aux, nullptr);
    for (int q = 0; q < n_aux; ++q) {
        if (!(p->used & (1u << q))) continue;
        hipEvent_t &ev = p->join[q];
        if (!ev && !(ev = new_event())) return COMBAT_ELAUNCH;
        if (hipEventRecord(ev, reinterpret_cast<hipStream_t>(aux_streams[q])) != hipSuccess ||
            hipStreamWaitEvent(main_st, ev, 0) != hipSuccess)
            return COMBAT_ELAUNCH;
    }
    p->used = 0;
    return COMBAT_OK;
}
