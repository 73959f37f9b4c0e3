export TMPDIR=/tmp
OUT=gpurun_out/r04_f; mkdir -p $OUT
python3 -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
python3 tools/flow_ab.py > $OUT/flow_ab.txt 2>&1; grep -v amdgpu.ids $OUT/flow_ab.txt
python3 tools/flow_ab.py --gops 1 --variants rel:0:0 rel:1:0 > $OUT/flow_ab_1gop.txt 2>&1; grep -v amdgpu.ids $OUT/flow_ab_1gop.txt
