# probes around the first packed add of the y chain: its operands and both halves of its result
def after(pat, new, start=0):
    k = next(k for k in range(start, len(K)) if pat in K[k])
    K[k + 1:k + 1] = new
    return k
kk = next(k for k, l in enumerate(K) if "v_pk_add_f32 v[36:37], v[36:37], v[4:5] op_sel:[0,1]" in l); b = max(k for k in range(kk) if K[k].startswith(".LBB"))
k = next(k for k in range(b, len(K)) if "v_pk_add_f32 v[36:37], v[36:37], v[4:5] op_sel:[0,1]" in K[k])
K[k + 1:k + 1] = ["\tv_mov_b32_e32 v73, v36", "\tv_mov_b32_e32 v74, v37"]
K[k:k] = ["\tv_mov_b32_e32 v70, v36", "\tv_mov_b32_e32 v71, v5", "\tv_mov_b32_e32 v72, v4"]
for old, new in (("v16, v26, s[2:3]", "v16, v70, s[2:3]"), ("v16, v27, s[6:7]", "v16, v71, s[6:7]"), ("v16, v45, s[10:11]", "v16, v72, s[10:11]"),
                 ("v16, v44, s[12:13]", "v16, v73, s[12:13]"), ("v16, v43, s[14:15]", "v16, v74, s[14:15]")):
    k = next(k for k, l in enumerate(K) if "global_store_dword " + old in l)
    K[k] = K[k].replace(old, new)
