for rep in 1 2; do
for L in prev new; do
  if [ $L = prev ]; then export NRX_LIB=$GRAFT_REPO_ROOT/news_recsys_amd/lib/libnrx_hip_prev.so; else unset NRX_LIB; fi
  echo "== $L"
  python tools/run_dcn2.py 320 400 bf16x3 2>&1 | grep "us "
  python tools/run_dcn2.py 320 400 bf16x3 train 2>&1 | grep "us "
  python tools/profile_dcn2_bwd.py 320 100 bf16x3 2>&1 | grep "us "
done; done
