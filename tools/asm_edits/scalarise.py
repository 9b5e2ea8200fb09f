# the two packed adds with op_sel:[0,1] replaced by pairs of plain v_add_f32 (same values)
for k, l in enumerate(K):
    if "v_pk_add_f32 v[36:37], v[36:37], v[4:5] op_sel:[0,1]" in l:
        K[k] = "\tv_add_f32_e32 v36, v36, v5\n\tv_add_f32_e32 v37, v37, v5"
    if "v_pk_add_f32 v[4:5], v[10:11], v[4:5] op_sel:[0,1]" in l:
        K[k] = "\tv_add_f32_e32 v4, v10, v5\n\tv_add_f32_e32 v5, v11, v5"
