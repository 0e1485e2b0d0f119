for tag in simple fixt; do
  cp tools/var/lib_$tag.so simurlacra_amd/csrc/libpysimhip.so
  echo "== $tag"; timeout -k 10 200 python -u tools/perf_envs.py 1048576 linear 0.0 identity 2>/dev/null | grep -E "QQ |OMO|BOB|QBB"
  timeout -k 10 200 python -u tools/perf_envs.py 1048576 linear 0.5 identity 2>/dev/null | grep -E "QQ |OMO|BOB|QBB"
done
