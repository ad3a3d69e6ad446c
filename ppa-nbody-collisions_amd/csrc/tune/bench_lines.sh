# The bench lines kept under profiles/ for a round: R=r03 bash ppa-nbody-collisions_amd/csrc/tune/bench_lines.sh   (on the GPU box)
O=gpurun_out
R=${R:-r03}
python bench.py > $O/${R}_bench.json 2> $O/${R}_bench.err &&
python bench.py --clean > $O/${R}_bench_clean.json 2>> $O/${R}_bench.err &&
python bench.py --force-comm --steps 20 --no-cpu-baseline > $O/${R}_bench_force_comm.json 2>> $O/${R}_bench.err &&
python bench.py --stock-radii --warmup 2 --steps 10 --no-cpu-baseline > $O/${R}_bench_c4_stock.json 2>> $O/${R}_bench.err &&
python bench.py --bodies 65536 --steps 50 --no-cpu-baseline > $O/${R}_bench_n65536.json 2>> $O/${R}_bench.err &&
python bench.py --bodies 65536 --stock-radii --warmup 0 --steps 50 --no-cpu-baseline > $O/${R}_bench_c3.json 2>> $O/${R}_bench.err &&
python bench.py --bodies 1048576 --fp64 --warmup 1 --steps 5 --cpu-budget 6 > $O/${R}_c5_bench.json 2>> $O/${R}_bench.err &&
python3 ppa-nbody-collisions_amd/csrc/tune/ref_launch_time.py 262144 6 > $O/${R}_reference_shaped_launch.txt 2>> $O/${R}_bench.err &&
python3 ppa-nbody-collisions_amd/csrc/tune/ref_launch_time.py 65536 20 >> $O/${R}_reference_shaped_launch.txt 2>> $O/${R}_bench.err
echo "bench_lines rc=$?"
