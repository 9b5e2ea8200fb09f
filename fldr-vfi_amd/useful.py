"""Configuration helper with the reference's name (useful.py:163-190)."""

_PAPER = dict(
    pcanet=True, mean_vector_norm=True, ds_normInput=True, scales=[8, 16, 32, 64], fractions=[4, 16, 64, 256],
    S_trn=3, S_tst=3, dataset="X4K1000FPS", oneEV=True, ref_feat_extrac=True, optimizeEV=True,
    lr_milestones=[70, 120, 170], ExacOneEV=True, takeBestModel=True, allImUp=True, softsplat=True,
    forwendflowloss=True, warp_alpha=0.05, sminterp=True, ownsmooth=True, noResidAddup=True, impmasksoftsplat=True,
    cutoffUnnec=True, fixsmoothtwistup=True, sminterpInpIm=True, patch_size=512, tempbottomflowfix=True,
)


def getmodelconfig(args):
    """Apply the --papermodel settings in place."""
    for k, v in _PAPER.items():
        setattr(args, k, list(v) if isinstance(v, list) else v)
    return args


class MyPWC():
    """PWC-Net teacher wrapper (useful.py:105-117).  Never constructed on the inference path
    (fLDRnet.py:56: mypwc = None); needs external weights that the reference does not ship."""

    def __init__(self, args, cpu=False, checkpoint='./OpticalFlow/pwc-checkpoint.pt'):
        import torch
        from OpticalFlow.PWCNet import PWCNet
        self.args = args
        self.flow_predictor = PWCNet()
        self.flow_predictor.load_state_dict(torch.load(checkpoint))
        if not cpu:
            self.flow_predictor.to(self.args.gpu)

    def get_flow(self, im0, im1):
        import torch
        flow = self.flow_predictor(torch.cat([im0, im1], dim=0), torch.cat([im1, im0], dim=0))
        flow01, flow10 = torch.split(flow, im0.shape[0], dim=0)
        return torch.cat([flow10, flow01], dim=1)
