for seed in 1 2 3; do
 for cfg in "f32:" "bf16x3:" "bf16x3:wgrad" "bf16x3:fwd" "bf16x3:dgrad" "bf16x3:fwd,dgrad"; do
  m=${cfg%%:*}; k=${cfg##*:}
  r=$(HEATNET_CONV_MATH=$m HEATNET_CONV_MATH_F32_KINDS=$k python tools/diag_grad.py 2 64 96 $seed 2>/dev/null | grep "median rel err")
  echo "seed $seed math=$m f32kinds=[$k] $r"
 done
done
