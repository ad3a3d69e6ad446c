P=ppa-nbody-collisions_amd
cp $P/libnbody_mi355x.so /tmp/orig.so
for v in S R2 R3 S R2 R3; do
  cp build/libnbody_$v.so $P/libnbody_mi355x.so
  echo "== $v"; python3 $P/csrc/tune/ring_probe.py 262144 8 58 3 2>&1 | grep -v amdgpu | grep "probe rank0\|variant=58" | cut -c1-220
done
cp /tmp/orig.so $P/libnbody_mi355x.so
