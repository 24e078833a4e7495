echo "base: $(python tools/quick_step.py 2>&1 | tail -n 1)"
for w in 96 160 192 256; do echo "wgs $w: $(COMBAT_WGRAD_WGS=$w python tools/quick_step.py 2>&1 | tail -n 1)"; done
for t in 1 2 4; do echo "main_tail $t: $(COMBAT_WGRAD_MAIN_TAIL=$t python tools/quick_step.py 2>&1 | tail -n 1)"; done
echo "defer: $(COMBAT_DEFER_REDUCE=1 python tools/quick_step.py 2>&1 | tail -n 1)"
echo "base: $(python tools/quick_step.py 2>&1 | tail -n 1)"
