# the same add with the register pair of src0 / dst the other way round is not expressible; instead: the failing packed add moved IN FRONT of the x chain
# (directly behind the branch target) — does the failure stay with the instruction or with the place?
kk = next(k for k, l in enumerate(K) if "v_pk_add_f32 v[36:37], v[36:37], v[4:5] op_sel:[0,1]" in l); b = max(k for k in range(kk) if K[k].startswith(".LBB"))
k = next(k for k in range(b, len(K)) if "v_pk_add_f32 v[36:37], v[36:37], v[4:5] op_sel:[0,1]" in K[k])
assert "v_mov_b32_e32 v37, v27" in K[k - 1] and "v_mov_b32_e32 v36, v33" in K[k - 2]
blk = K[k - 2:k + 1]
# v[36:37] is a temporary of the x chain in between: use v[70:71] for the early copy and move it into place where the original stood
new = [x.replace("v36", "v70").replace("v37", "v71").replace("v[36:37]", "v[70:71]") for x in blk]
K[k - 2:k + 1] = ["\tv_mov_b32_e32 v36, v70", "\tv_mov_b32_e32 v37, v71"]
K[b + 1:b + 1] = new
