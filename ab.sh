mkdir -p gpurun_out/r4
python mofreak_amd/tools/ab_tile.py mofreak_amd/libmofreak_hip.so mofreak_amd/_exp/libvar_abl1.so mofreak_amd/_exp/libvar_abl4.so mofreak_amd/_exp/libvar_abl5.so mofreak_amd/libmofreak_hip.so > gpurun_out/r4/stage_ablation.log 2>&1; echo rc=$?
cat gpurun_out/r4/stage_ablation.log | tail -12
