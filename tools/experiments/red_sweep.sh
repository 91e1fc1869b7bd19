for v in "" red2048 red1024; do
  if [ -n "$v" ]; then export ATX_LIBRARY=$GRAFT_REPO_ROOT/anemoi-transform_amd/lib/variants/libatx_$v.so; fi
  echo "== ${v:-head 8192}"
  python tools/small_case_bench.py 2>&1 | grep "reduce.*us per call"
  python tools/kernel_bench.py 2>&1 | grep -i "reduce" | cut -c1-100
done
