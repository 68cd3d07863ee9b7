for v in "A::" "B:2:" "C::-DSA_WAVE_INLINE_CALLBACKS" "D:2:-DSA_WAVE_INLINE_CALLBACKS"; do IFS=: read tag eu defs <<< "$v"
  if [ -n "$eu" ]; then export SA_WAVES_PER_EU=$eu; else unset SA_WAVES_PER_EU; fi
  if [ -n "$defs" ]; then export SA_KERNEL_DEFINES="$defs"; else unset SA_KERNEL_DEFINES; fi
  echo "variant $tag eu=$eu defs=$defs"; timeout 300 python tools/bench_configs.py seir 16384 2>&1 | tail -1
done
