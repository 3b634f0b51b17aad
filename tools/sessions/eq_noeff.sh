for v in "" ${1:-PF2}; do
  if [ -n "$v" ]; then export DC_DDIM_LIB="$PWD/diffusion-conductor_amd/libdc_ddim_$v.alt"; else unset DC_DDIM_LIB; fi
  DC_NO_EFF=1 python tools/ab_equal.py 4
  DC_NO_EFF=1 DC_RAGGED=1 python tools/ab_equal.py 5
done
