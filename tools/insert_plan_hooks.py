"""One-off source edit: give every stream-taking extern "C" entry point its COMBAT_PLAN_HOOK line (csrc/plan.hpp).
Idempotent; run from the repo root after adding an entry point:  python tools/insert_plan_hooks.py"""
import glob
import re

PAT = re.compile(r'(extern "C" int (combat_\w+)\(([^)]*)\)\s*\{\n)(?!\s*COMBAT_PLAN_HOOK)', re.S)
total = 0
for path in sorted(glob.glob("combat_amd/csrc/*.hip")):
    src = open(path).read()

    def repl(m):
        global total
        name, args = m.group(2), [a.strip() for a in m.group(3).replace("\n", " ").split(",")]
        names = [re.split(r"[\s\*]+", a)[-1] for a in args]
        if names[-1] != "stream":
            return m.group(0)
        total += 1
        return m.group(1) + "    COMBAT_PLAN_HOOK(%s, %s);\n" % (name, ", ".join(names[:-1]))

    new = PAT.sub(repl, src)
    if new != src:
        if '#include "plan.hpp"' not in new:
            new = re.sub(r'(#include "[\w_]+\.hpp"\n)', r'\1#include "plan.hpp"\n', new, count=1)
        open(path, "w").write(new)
print("hooks inserted:", total)
