# Item 2(b) of the round-5 review: the 16 DCN layers at 0.5 / 1 / 2 px offsets through (a) the default routing, (b) the column-buffer
# path on every eligible geometry, (c) the generic three-pass kernels.   bash tools/route_probe.sh > gpurun_out/route_probe.txt
cd $GRAFT_REPO_ROOT
for s in 0.5 1.0 2.0; do
  echo "== default routing, offsets $s px"; python tools/time_dcn_layers.py 8 f32 $s
  echo "== DCD_DCN_DENSE=1, offsets $s px"; DCD_DCN_DENSE=1 python tools/time_dcn_layers.py 8 f32 $s
done
echo "== DCD_BWD_SWEEP=0 (generic three-pass), offsets 2 px"; DCD_BWD_SWEEP=0 python tools/time_dcn_layers.py 8 f32 2.0
echo "== DCD_DCN_HANDOVER=never, offsets 2 px"; DCD_DCN_HANDOVER=never python tools/time_dcn_layers.py 8 f32 2.0
echo "== DCD_DCN_DENSE=1 DCD_BWD_SWEEP=0, offsets 2 px"; DCD_DCN_DENSE=1 DCD_BWD_SWEEP=0 python tools/time_dcn_layers.py 8 f32 2.0
