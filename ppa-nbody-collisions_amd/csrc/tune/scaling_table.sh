# Kernel-level strong scaling of the force kernel at N=262144 on ONE GPU: one rank of a G-rank partition, launched back
# to back in steady state (rank_kernel.py -> nbody_debug_force_only), for G = 1, 2, 4, 8, same box.
P=ppa-nbody-collisions_amd/csrc/tune/rank_kernel.py
for g in 1 2 4 8; do python3 $P 262144 $g $((g/2)) 0 $((6*g)); done
for g in 8 4 2 1; do python3 $P 262144 $g $((g/2)) 0 $((6*g)); done     # and back: the boxes' clocks drift
python3 $P 262144 1 0 31 6
python3 $P 65536 1 0 0 60
python3 $P 65536 1 0 0 60 stock
python3 $P 65536 1 0 31 60
