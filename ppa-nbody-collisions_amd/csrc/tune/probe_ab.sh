# In-kernel phase stamps (kernel_variant 58, ring_probe.py) of several builds of the library on the same box:
#   VARIANTS="A B" SHAPE="262144 8" bash probe_ab.sh      (build/libnbody_A.so, build/libnbody_B.so; build/ travels with gpurun)
P=ppa-nbody-collisions_amd
cp $P/libnbody_mi355x.so /tmp/orig.so
for v in ${VARIANTS:-A B} ${VARIANTS:-A B}; do
  cp build/libnbody_$v.so $P/libnbody_mi355x.so
  echo "== $v"; python3 $P/csrc/tune/ring_probe.py ${SHAPE:-262144 8} 58 3 2>&1 | grep -v amdgpu | grep "probe rank0\|variant=58" | cut -c1-220
done
cp /tmp/orig.so $P/libnbody_mi355x.so
