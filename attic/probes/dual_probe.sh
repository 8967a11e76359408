# needs a probe build: tools/build_exp.sh gemm1x1.hip G1_DUAL_EXP 1 (POCO_G1_DUAL is ignored by the product build)
for cfg in 1,1,1 1,1,3 1,1,4 1,1,5 1,1,6 2,2,1 2,2,4 2,2,5 2,2,6 1,4,1 1,4,5 1,4,6 2,1,5 4,1,5 1,2,5 1,2,6; do
  echo -n "$cfg: "; POCO_G1_DUAL=$cfg python tools/shape_report.py --variant resnet50-cliff --batch 64 2>&1 | grep "'other', 15\|^total" | tr '\n' ' '; echo
done
