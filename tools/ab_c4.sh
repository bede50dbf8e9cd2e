for flags in "$@"; do
  KM_EXTRA_FLAGS="$flags" python -m koemorph_amd.build --force > /dev/null 2>&1 && echo "$flags" && B=256 python tools/bench_c4.py 2>/dev/null | head -1 | cut -c1-200
done
