# PC sampling probe (round 5): is it supported here, and where do the H = 32 backward's waves sit?
out=gpurun_out/r05_pcs; mkdir -p $out
timeout 60 rocprofv3-avail list --pc-sampling > $out/avail.txt 2>&1; echo "avail rc=$?" >> $out/avail.txt
timeout 60 rocprofv3-avail info --pc-sampling >> $out/avail.txt 2>&1
cat $out/avail.txt | tail -30
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method stochastic --pc-sampling-unit cycles --pc-sampling-interval 1048576 --kernel-trace -d $GRAFT_REPO_ROOT/$out/stoch -o pcs --output-format csv json -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/$out/stoch.log 2>&1
echo "stochastic rc=$?"; tail -5 $GRAFT_REPO_ROOT/$out/stoch.log
ls -la $GRAFT_REPO_ROOT/$out/stoch/* 2>/dev/null | head
