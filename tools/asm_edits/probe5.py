# where the failing waves run: HW_ID and XCC_ID through the flowback_0 planes, the packed add's low result through im0_tot[0]
kk = next(k for k, l in enumerate(K) if "v_pk_add_f32 v[36:37], v[36:37], v[4:5] op_sel:[0,1]" in l); b = max(k for k in range(kk) if K[k].startswith(".LBB"))
k = next(k for k in range(b, len(K)) if "v_pk_add_f32 v[36:37], v[36:37], v[4:5] op_sel:[0,1]" in K[k])
K[k + 1:k + 1] = ["\tv_mov_b32_e32 v72, v36"]
K[k:k] = ["\ts_getreg_b32 s90, hwreg(HW_REG_HW_ID)", "\ts_getreg_b32 s91, hwreg(HW_REG_XCC_ID)", "\ts_nop 3", "\tv_mov_b32_e32 v70, s90", "\tv_mov_b32_e32 v71, s91"]
for old, new in (("v16, v26, s[2:3]", "v16, v70, s[2:3]"), ("v16, v27, s[6:7]", "v16, v71, s[6:7]"), ("v16, v45, s[10:11]", "v16, v72, s[10:11]")):
    k = next(k for k, l in enumerate(K) if "global_store_dword " + old in l)
    K[k] = K[k].replace(old, new)
