# Quick parity pass of an alternative build of the library (build/libnbody_$1.so) on the GPU box: the golden free runs of the
# ring-kernel variants, the event sets, the random sweeps, the collision-screen and sharded cases.  The in-tree library is
# put back afterwards.  Usage: bash ab_parity.sh E
P=ppa-nbody-collisions_amd
cp $P/libnbody_mi355x.so /tmp/orig_parity.so
cp build/libnbody_$1.so $P/libnbody_mi355x.so
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "golden_free_run or events_match or random_ or collision_screen or sharded or unbounded or extreme or clean_semantics or big_golden or ragged or reference_shaped or handoff" 2>&1 | tail -4
rc=$?
cp /tmp/orig_parity.so $P/libnbody_mi355x.so
exit $rc
