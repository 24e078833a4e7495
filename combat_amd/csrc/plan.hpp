// Recording hook of the C-side plan replayer (plan.cpp).  Every stream-taking entry point of the C ABI starts with
// COMBAT_PLAN_HOOK(name, args...): while the calling thread has armed a recording (combat_plan_record), the call is
// captured -- its arguments by value, pointers as pointers -- into the plan instead of being launched, and replayed
// later with the stream the plan runs on.
#pragma once
#include <functional>

namespace combat_plan_detail {
bool armed();
int capture(std::function<int(void *)> call);
}  // namespace combat_plan_detail

#define COMBAT_PLAN_HOOK(fn, ...)                                                                        \
    do {                                                                                                 \
        if (__builtin_expect(combat_plan_detail::armed(), 0))                                            \
            return combat_plan_detail::capture([=](void *stream_) { return fn(__VA_ARGS__, stream_); }); \
    } while (0)
