# Every kernel-selection switch as a short bs-256 bench run (the option-equivalence TEST runs at N = 4: a kernel generation that only breaks at the full
# batch - as the packed depthwise weight gradient DWW = 6 did, found and removed in round 6 - shows up here). usage (GPU box): bash tools/option_smoke.sh [out file]
O=${1:-gpurun_out/option_smoke.txt}
: > $O
for o in "" "DW=6" "DW=5" "DW=3" "DWW=5" "TN=1" "TN3_BLOCKS=0" "TN3_BLOCKS=256" "TNG_BLOCKS=0" "NT_GLDS=0" "NT_GLDS64=0,NT_BK32=0" "CS_SPLIT=0" "RSC_PF=0,rsc_small=0" "rsc_small=0" "RSC_N40=1,RSC_N80=0" \
         "RSC_W5=0" "RSC_ATOMIC=320" "RSC1=0" "RSC1=2,RSC1_ATOMIC=0" "RSP=0" "RSP=2,RSP_NARROW=15" "RSP_NWV=8,RSP_NARROW=15" "RSN3=0" "RSN3=4" "NT_RING=0" "NT_RING=464" "RST_NW=4" "FOLD_GROUP=1" "EVX=0" "DET=1,det=1" \
         "stats_wgrad=0" "wg_fused=0" "down_fused=0" "ps=0" "ps=3" "ps_xcd_barrier=0" "lanes=0" "dzr=0" "grn_fold=0" "rsc=0" "z_free=0" "loss_onepass=0" "stem_front=0" "heads_merged=0" "wgrad_group=0" "dw_group=9"; do
  r=$(MPMAE_ENGINE_OPTS="$o" timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json
try:
    d=json.loads(sys.stdin.read()); print(d['ms_per_step'], 'final_loss', d['config']['final_loss'])
except Exception as e:
    print('FAILED')")
  echo "$o: $r" | tee -a $O
done
