for mb in 0 1 2 4 8; do
  echo "== DCD_BI_MB=$mb"
  DCD_BI_MB=$mb python tools/time_dcn_layers.py 8 f32 0.5 | grep -E "256|512|128"
done
for mb in 0 1 2 4 8; do
  echo "== B=1 DCD_BI_MB=$mb"
  DCD_BI_MB=$mb python tools/time_dcn_layers.py 1 f32 0.5 | grep -E "TOTAL"
done
