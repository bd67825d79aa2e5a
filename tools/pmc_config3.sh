set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
WC="python3 tools/bench_ops.py warp_corr --size 512 --batch 32 --levels 3 --variants 0"
CV="python3 tools/bench_ops.py conv --variants 0 --filter L1_R"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc3_fetch -- $WC > gpurun_out/pmc3_a.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc3_write -- $WC > gpurun_out/pmc3_b.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/pmc3_tcc -- $WC > gpurun_out/pmc3_c.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc3_mfma -- python3 tools/bench_ops.py conv --variants 0 --filter "L1 R.conv_R.2" > gpurun_out/pmc3_d.log 2>&1
echo ok
