# probes of tap 1 of the image path (packed odd halves) through the flowback_0 / im0_tot stores: v70.. = copies made where the value is produced
def after(pat, new, start=0):
    k = next(k for k in range(start, len(K)) if pat in K[k])
    K[k + 1:k + 1] = new
    return k
kk = next(k for k, l in enumerate(K) if "v_pk_add_f32 v[36:37], v[36:37], v[4:5] op_sel:[0,1]" in l); b = max(k for k in range(kk) if K[k].startswith(".LBB"))
after("v_cvt_i32_f32_e32 v46, v21", ["\tv_cvt_f32_i32_e32 v72, v46"], b)
after("v_cvt_i32_f32_e32 v51, v29", ["\tv_cvt_f32_i32_e32 v73, v51"], b)
after("v_pk_mul_f32 v[46:47], v[38:39], v[40:41]", ["\tv_mov_b32_e32 v74, v47"], b)
after("v_cndmask_b32_e64 v41, 0, v47, s[18:19]", ["\tv_mov_b32_e32 v71, v41"], b)
k = next(k for k in range(b, len(K)) if "v_cmp_gt_f32_e32 vcc, s82, v43" in K[k])
K[k:k] = ["\tv_mov_b32_e32 v70, v43"]
# the stores: flowback_0 (v26, v27) and im0_tot (v45, v44, v43)
for old, new in (("v16, v26, s[2:3]", "v16, v70, s[2:3]"), ("v16, v27, s[6:7]", "v16, v71, s[6:7]"), ("v16, v45, s[10:11]", "v16, v72, s[10:11]"),
                 ("v16, v44, s[12:13]", "v16, v73, s[12:13]"), ("v16, v43, s[14:15]", "v16, v74, s[14:15]")):
    k = next(k for k, l in enumerate(K) if "global_store_dword " + old in l)
    K[k] = K[k].replace(old, new)
