cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04b; mkdir -p $O
for m in 0 1 2; do DS6G_BG_WXCD=$m python tools/bench_bwgrad.py > $O/t_wxcd$m.txt 2>&1 || exit 1; done
for m in 0 1 2; do
  DS6G_BG_WXCD=$m REPS=2 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc$m -o p -- python tools/bench_bwgrad.py > $O/pmc$m.log 2>&1 || exit 1
  python tools/bwgrad_traffic.py $O/pmc$m/p_counter_collection.csv > $O/traffic_wxcd$m.txt || exit 1
  rm -rf $O/pmc$m
done
paste $O/t_wxcd0.txt $O/t_wxcd1.txt $O/t_wxcd2.txt | cut -c1-260
