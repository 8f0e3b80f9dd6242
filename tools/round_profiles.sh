#!/bin/bash
# The round's evidence in one go: bench line + rocprofv3 kernel statistics of the same command,
# counter passes (HBM bytes, VALU / LDS / MFMA counters) per kernel, kernel statistics at a rank's
# share of the items.  usage: tools/round_profiles.sh TAG   (writes gpurun_out/TAG/...)
tag=$1
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
bash tools/pmc_collect.sh gpurun_out/$tag/pmc || exit 1
python3 tools/pmc_report.py gpurun_out/$tag/pmc gpurun_out/$tag/pmc_kernels.json config:n=10000,m=4096 \
  "ess_f_kernel=ess_f" "ess_param_kernel=ess_param" "theta_eval_bucket=theta_eval_bucket" "theta_table_kernel=theta_table" \
  "gemm_direct_kernel<false, 5>=nu_rest" "gemm_direct_lone_kernel<false, 2>=nu_chunks" "gemm_direct_kernel<false, 4>=nu_whole" \
  "gemm_direct_kernel<true, 1>=syrk" "gemm_direct_lone_kernel<true, 1>=syrk_lookahead" "potrf_diag_kernel=potrf_diag" \
  "gemm_direct_small_kernel<true>=gemm_small" "trsm_panel_kernel=trsm_panel" \
  "potrf_diag_tall_kernel=potrf_diag_tall" "potrf_inpanel_kernel=potrf_inpanel" "potrf_tall_step_kernel=potrf_tall_step" \
  "potrf_rawcopy_kernel=potrf_rawcopy" "theta_decide_kernel=theta_decide" "theta_scatter_kernel=theta_scatter" "fill_normal_kernel=normal_draws" "gemm_direct_lone_kernel<false, 0>=nu_tail_panels" > /dev/null || exit 1
# the bench line quotes the counters' HBM bytes from the committed file: put this run's in place first
cp gpurun_out/$tag/pmc_kernels.json profiles/$(python3 -c "import bench; print(bench.PMC_FILE)") || exit 1
bash tools/profile_round.sh gpurun_out/$tag/full || exit 1
BENCH_ARGS="--m 1024" bash tools/kstats.sh gpurun_out/$tag/m1024 > gpurun_out/$tag/m1024.txt || exit 1
BENCH_ARGS="--m 512" bash tools/kstats.sh gpurun_out/$tag/m512 > gpurun_out/$tag/m512.txt || exit 1
# the bench line itself at a rank's share of the items (2048 / 1024 / 512: two, four, eight ranks)
for M in 2048 1024 512; do
  timeout -k 10 300 python -u bench.py --steps 20 --warmup 3 --m $M --no-cpu-baseline --no-variants > gpurun_out/$tag/items$M.json 2> gpurun_out/$tag/items$M.err || { tail -5 gpurun_out/$tag/items$M.err; exit 1; }
done
echo done
