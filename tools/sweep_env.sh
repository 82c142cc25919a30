run() { env "$@" python bench.py --steps 40 --warmup 8 --no-cpu-baseline 2>&1 | tail -1 | sed 's/.*"ms_per_step": \([0-9.]*\).*/\1/'; }
echo "base $(run A=1) $(run A=1)"
for v in 256 384 768 1024; do echo "TN_BLOCKS=$v $(run MPMAE_TN_BLOCKS=$v)"; done
for v in 128 384; do echo "TN_BLOCKS_BIG=$v $(run MPMAE_TN_BLOCKS_BIG=$v)"; done
for v in 128 512; do echo "TN_MINROWS=$v $(run MPMAE_TN_MINROWS=$v)"; done
for v in 1024 2048 3072; do echo "RSC_BLOCKS=$v $(run MPMAE_RSC_BLOCKS=$v)"; done
for v in 256 384 448; do echo "DW6_T8=$v $(run MPMAE_DW6_T8=$v)"; done
for v in 256 384; do echo "DW6_T4=$v $(run MPMAE_DW6_T4=$v)"; done
for v in 256 384; do echo "DW6_T2=$v $(run MPMAE_DW6_T2=$v)"; done
echo "base $(run A=1)"
for v in 256 1024; do echo "STB_BLOCKS=$v $(run MPMAE_STB_BLOCKS=$v)"; done
for v in 2 3; do echo "RING=$v $(run MPMAE_RING=$v)"; done
echo "DWW_NB: $(grep -c . /dev/null)"
