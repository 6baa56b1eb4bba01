F="--no-cpu-baseline --no-host-path --no-other-configs"
for m in 1 0; do echo merged=$m; JCH_LV_MERGED=$m JCH_LV_DEBUG=1 python bench.py --steps 1 --warmup 1 $F 2>&1 >/dev/null | grep "call 1 \|call 12\|call 24" | head -3; done
