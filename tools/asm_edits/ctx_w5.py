# context variants around the failing packed add: 1 every outstanding memory operation waited for at the block's entry, 2 128 idle cycles in front of
# the instruction, 3 src1 from a fresh copy v[70:71] of v[4:5] (other register banks), 5 src1 from a copy in v[72:73] whose LOW register holds 0.0
V = 5
kk = next(k for k, l in enumerate(K) if "v_pk_add_f32 v[36:37], v[36:37], v[4:5] op_sel:[0,1]" in l); b = max(k for k in range(kk) if K[k].startswith(".LBB"))
k = next(k for k in range(b, len(K)) if "v_pk_add_f32 v[36:37], v[36:37], v[4:5] op_sel:[0,1]" in K[k])
if V == 1: K[b + 1:b + 1] = ["\ts_waitcnt vmcnt(0) lgkmcnt(0)", "\ts_nop 7"]
if V == 2: K[k:k] = ["\ts_sleep 2"]
if V == 3:
    K[k] = "\tv_pk_add_f32 v[36:37], v[36:37], v[70:71] op_sel:[0,1]"
    K[b + 1:b + 1] = ["\tv_mov_b32_e32 v70, v4", "\tv_mov_b32_e32 v71, v5"]
if V == 5:
    K[k] = "\tv_pk_add_f32 v[36:37], v[36:37], v[72:73] op_sel:[0,1]"
    K[b + 1:b + 1] = ["\tv_mov_b32_e32 v72, 0", "\tv_mov_b32_e32 v73, v5"]
