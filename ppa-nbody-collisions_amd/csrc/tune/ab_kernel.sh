# Same-box A/B of builds of the library (build/libnbody_A.so, build/libnbody_B.so, ...; VARIANTS="A B C": build/ is git-ignored but travels with
# gpurun): one rank's force kernel in steady state (rank_kernel.py) at the 1-rank, 8-rank and N=65536 shapes, two rounds.
P=ppa-nbody-collisions_amd
# shapes = rank_kernel.py arguments "N G rank variant reps [stock]"; SHAPE_LIST="a;b;c" overrides the default set
if [ -n "$SHAPE_LIST" ]; then IFS=';' read -ra SHAPES <<< "$SHAPE_LIST"; else
  SHAPES=("262144 1 0 0 6" "262144 8 4 0 40" "65536 1 0 0 60" "65536 1 0 0 60 stock"); fi
cp $P/libnbody_mi355x.so /tmp/orig.so
for round in 1 2; do
  for v in ${VARIANTS:-A B}; do
    cp build/libnbody_$v.so $P/libnbody_mi355x.so
    for shape in "${SHAPES[@]}"; do
      echo -n "$v r$round: "; python3 $P/csrc/tune/rank_kernel.py $shape 2>&1 | grep -v amdgpu | cut -c1-120
    done
  done
done
cp /tmp/orig.so $P/libnbody_mi355x.so
