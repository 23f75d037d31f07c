for cfg in C2 C3 C5 HBM32; do
  for v in default plain default plain; do
    if [ $v = plain ]; then export KMC_LIB_PATH=$PWD/kissmcmc.jl_amd/libkmc_var_plain.so; else unset KMC_LIB_PATH; fi
    G=2048; [ $cfg = C5 ] && G=512; [ $cfg = HBM32 ] && G=200
    echo -n "$cfg $v: "; python scripts/run_cfg.py $cfg $G 1 2>&1 | grep "us/half-step" | sed 's/.*us\/half-step//'
  done
done
