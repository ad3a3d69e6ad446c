# rocprofv3 passes behind profiles/: kernel stats of the default bench command, FETCH_SIZE / WRITE_SIZE in separate
# passes (MI355X_MICROARCH.md, HBM section), and the SQ counters of the ring kernel at three shapes.  Run on the GPU box:
#   R=r02 bash ppa-nbody-collisions_amd/csrc/tune/profile_round.sh   (outputs under gpurun_out/)
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
# (a) kernel trace + stats of the default bench command
rocprofv3 --kernel-trace --stats -d $O/${R:-r02}_prof_bench -o run --output-format csv -- python3 bench.py --steps 10 --warmup 2 > $O/${R:-r02}_prof_bench.json 2> $O/${R:-r02}_prof_bench.err
# (b) HBM-side traffic: separate PMC passes
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/${R:-r02}_pmc_fetch -o runc --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/${R:-r02}_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/${R:-r02}_pmc_write -o runc --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/${R:-r02}_pmc_write.log 2>&1
# (c) SQ counters of the ring kernel: one rank of 8 (2x8 workgroups), N=65536 on one GPU (4x4), N=262144 one GPU (4x4)
P=ppa-nbody-collisions_amd/csrc/tune/rank_kernel.py
for shape in "262144 8 3 0 4 g8" "65536 1 0 0 8 n64k" "262144 1 0 0 2 g1"; do
  set -- $shape
  rocprofv3 --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_LDS SQ_INST_LEVEL_LDS SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_WAIT_INST_LDS SQ_WAVE_CYCLES --kernel-trace -d $O/${R:-r02}_pmc_ring_${6}_x1 -o runc --output-format csv -- python3 $P $1 $2 $3 $4 $5 > $O/${R:-r02}_pmc_ring_${6}_x1.log 2>&1
  rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace -d $O/${R:-r02}_pmc_ring_${6}_x2 -o runc --output-format csv -- python3 $P $1 $2 $3 $4 $5 > $O/${R:-r02}_pmc_ring_${6}_x2.log 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_TRANS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace -d $O/${R:-r02}_pmc_ring_${6}_x3 -o runc --output-format csv -- python3 $P $1 $2 $3 $4 $5 > $O/${R:-r02}_pmc_ring_${6}_x3.log 2>&1
done
ls $O | grep ${R:-r02}_
# (d) the fp64 production kernel at C5's size on one GPU and at C5's 8-rank shape
for shape in "1048576 1 0 0 2 c5g1" "1048576 8 4 0 6 c5g8"; do
  set -- $shape
  rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS --kernel-trace -d $O/${R:-r02}_pmc_f64_${6}_x1 -o runc --output-format csv -- python3 $P $1 $2 $3 $4 $5 r0 fp64 > $O/${R:-r02}_pmc_f64_${6}_x1.log 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_TRANS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace -d $O/${R:-r02}_pmc_f64_${6}_x2 -o runc --output-format csv -- python3 $P $1 $2 $3 $4 $5 r0 fp64 > $O/${R:-r02}_pmc_f64_${6}_x2.log 2>&1
done
