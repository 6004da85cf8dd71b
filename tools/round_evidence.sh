#!/bin/bash
# usage (GPU box): tools/round_evidence.sh <tag> [head]   -> gpurun_out/<tag>_*: everything the round's profiles/ entries are copied from
#   (head: `git rev-parse --short HEAD` of the code commit, stamped into every bench line as "head" -- the box has no .git)
#   <tag>_bench_full.json            the driver-form bench line (python bench.py, default flags)
#   <tag>_{headline,224,bu64,pre,wrn,bert,hubert}.stats.txt + _bench_under_rocprof.json   rocprofv3 --kernel-trace --stats of the workloads
#   <tag>_roofline_vs_rocprof.txt   the bench line's live per-launch time of the dominant kernel against the rocprofv3 average (must agree within 3 %)
#   <tag>_hbm_traffic.json           FETCH_SIZE / WRITE_SIZE passes of the headline (tools/traffic.sh)
#   <tag>_pmc{1,2}.pmc.txt           SQ counters of the headline (tools/pmc.sh)
tag=$1
export SR_BENCH_HEAD=$2
cd $GRAFT_REPO_ROOT
# HBM traffic first: the bench line reads profiles/<tag>_hbm_traffic.json (static, labelled as such) for roofline.traffic
bash tools/traffic.sh ${tag} --no-also --repeats 1
cp gpurun_out/${tag}_hbm_traffic.json profiles/${tag}_hbm_traffic.json
python bench.py 2> gpurun_out/${tag}_bench_full.err | tail -1 > gpurun_out/${tag}_bench_full.json
Q="--steps 20 --warmup 5 --repeats 1 --no-cpu-baseline --no-roofline --no-also"
bash tools/prof.sh ${tag}_headline $Q; grep -h "^{\"metric\"" gpurun_out/${tag}_headline.log > gpurun_out/${tag}_headline_bench_under_rocprof.json
# the roofline pass under the profiler: the SAME launches timed by bench.py (dispatch-bound events) and by rocprofv3 must agree within 3 %
bash tools/prof.sh ${tag}_roofcheck --steps 20 --warmup 5 --repeats 1 --no-cpu-baseline --no-also
grep -h "^{\"metric\"" gpurun_out/${tag}_roofcheck.log > gpurun_out/${tag}_roofcheck_bench_under_rocprof.json
python tools/check_roofline_vs_rocprof.py gpurun_out/${tag}_roofcheck_bench_under_rocprof.json gpurun_out/${tag}_roofcheck.stats.txt gpurun_out/${tag}_roofcheck.durations.json \
  > gpurun_out/${tag}_roofline_vs_rocprof.txt 2>&1 || echo "ROOFLINE CHECK FAILED (see gpurun_out/${tag}_roofline_vs_rocprof.txt)"
echo "--- the unprofiled driver-form line against the profile of the plain run (different processes: the profiler slows the step) ---" >> gpurun_out/${tag}_roofline_vs_rocprof.txt
python tools/check_roofline_vs_rocprof.py gpurun_out/${tag}_bench_full.json gpurun_out/${tag}_headline.stats.txt 0.10 >> gpurun_out/${tag}_roofline_vs_rocprof.txt 2>&1
cat gpurun_out/${tag}_roofline_vs_rocprof.txt
bash tools/prof.sh ${tag}_224 $Q --img 224; grep -h "^{\"metric\"" gpurun_out/${tag}_224.log > gpurun_out/${tag}_224_bench_under_rocprof.json
bash tools/prof.sh ${tag}_bu64 $Q --bu 64 --steps 6 --warmup 2; grep -h "^{\"metric\"" gpurun_out/${tag}_bu64.log > gpurun_out/${tag}_bu64_bench_under_rocprof.json
bash tools/prof.sh ${tag}_pre $Q --regime pre; grep -h "^{\"metric\"" gpurun_out/${tag}_pre.log > gpurun_out/${tag}_pre_bench_under_rocprof.json
bash tools/prof.sh ${tag}_wrn $Q --net wrn --bu 64 --steps 8 --warmup 3; grep -h "^{\"metric\"" gpurun_out/${tag}_wrn.log > gpurun_out/${tag}_wrn_bench_under_rocprof.json
bash tools/prof.sh ${tag}_bert $Q --net bert --steps 4 --warmup 2; grep -h "^{\"metric\"" gpurun_out/${tag}_bert.log > gpurun_out/${tag}_bert_bench_under_rocprof.json
bash tools/prof.sh ${tag}_hubert $Q --net hubert --steps 4 --warmup 2; grep -h "^{\"metric\"" gpurun_out/${tag}_hubert.log > gpurun_out/${tag}_hubert_bench_under_rocprof.json
# data parallel on the live backend with one forced rank (1-rank RCCL communicator): the line with rccl_ranks 1 + every pinned exchange, and the
# collective-by-collective check
python bench.py --force-dp --no-also --no-cpu-baseline --no-roofline 2> gpurun_out/${tag}_force_dp_rccl.err | grep "^{" | tail -1 > gpurun_out/${tag}_force_dp_rccl.json
python tools/rccl_one_rank_check.py 2>&1 | grep -v "amdgpu.ids\|socket.cpp" | cut -c1-2000 > gpurun_out/${tag}_rccl_one_rank.txt
SR_PHASES=1 python bench.py --no-also --no-cpu-baseline --no-roofline --repeats 3 2>&1 | grep -i "^phases" > gpurun_out/${tag}_phases.txt
SR_PHASES=1 python bench.py --no-also --no-cpu-baseline --no-roofline --repeats 3 --regime pre 2>&1 | grep -i "^phases" >> gpurun_out/${tag}_phases.txt
P="$GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline --no-roofline --no-also"
bash tools/pmc.sh ${tag}_pmc1 "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_LDS_IDX_ACTIVE" $P
bash tools/pmc.sh ${tag}_pmc2 "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" $P
rm -f gpurun_out/${tag}_*.dispatch.txt gpurun_out/${tag}_[a-z0-9]*.durations.json.bak; for f in gpurun_out/${tag}_*.durations.json; do case $f in *roofcheck*) ;; *) rm -f $f;; esac; done
ls -la gpurun_out | grep ${tag}
