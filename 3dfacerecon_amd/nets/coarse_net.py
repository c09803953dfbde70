"""The callers of the hot path in BASELINE.json configs 3-5: CoarseNet (iterated parameter regressor) and FineNet, as
plain torch.nn modules (SURVEY.md 8f rank 3 -- stock conv nets; no custom kernels here, no torchvision dependency).

Architecture facts taken from the reference (nets/network.py):
  * build_coarse_net (103-136): per iteration a SEPARATE net: 7-channel input [maskimg | pncc | normal] (122) ->
    conv 7x7 / 2 -> 64 (124) -> four slim resnet_v1 bottleneck blocks of 2 units each, base depths 64/128/256/512,
    block strides 1/2/2/2 with the stride on the block's last unit (125-128) -> conv 3x3 -> 235 (131) -> global average
    pool (132) -> fully connected 235 without norm / activation (133) -> set_constraints (136).  Conv layers carry
    batch-norm + ReLU (arg_scope at 76-83).
  * build_fine_net (311-333): input [im_gray | coarse depth] -> conv1 x2 (64) -> pool -> conv2 x2 (128) -> pool ->
    conv3 x3 (256); hypercolumn of conv1, upconv2 (2x2/2 -> 128) and upconv3 (two 2x2/2 -> 256); 1x1 convs 50, 50, 10
    and a linear 1x1 -> 1 depth map.
The render loop between the iterations is the MI355X hot path: FaceRecNet.vertices_transform (fr_decode_3dmm) and
FaceRecNet.coarse_net_input (fr_rendering_layer_forward), both differentiable.
"""
import os
import warnings

import torch
import torch.nn as nn

# MIOpen 3.5.0 on gfx950: while PyTorch's default (non-immediate) convolution path benchmarks the applicable solvers the
# first time it sees a shape, the assembly implicit-GEMM backward-data kernel `igemm_bwd_gtcx35_nhwc_fp32_bx0_ex1_bt256x64x4_...`
# (solver ConvAsmImplicitGemmGTCDynamicBwdXdlopsNHWC) reads past the end of one of its buffers on this model's backward
# shapes.  Inside torch's caching allocator the overrun usually lands in mapped memory and goes unnoticed; when the buffer
# ends on the last page of a segment the process dies with "Memory access fault by GPU" (found with serialized launches +
# AMD_LOG_LEVEL=3 on a fresh box with an empty MIOpen user database: the faulting launch is that kernel, inside
# miopenFindConvolutionBackwardDataAlgorithm; profiles/round4_probes/r4r).
# The cure is a PROCESS setting (it switches the solver off for every convolution of the host application), so this library
# module does not apply it behind the caller's back: the entry points (examples/coarse_loop.py, tests/conftest.py) call
# apply_miopen_workaround() before the first convolution runs -- MIOpen reads the variable when it first checks the solver's
# applicability -- INTEGRATION.md lists it as a launch requirement, and building a trainable net without it warns.
MIOPEN_WORKAROUND = ("MIOPEN_DEBUG_CONV_IMPLICIT_GEMM_ASM_BWD_GTC_XDLOPS_NHWC", "0")


def apply_miopen_workaround():
    """For a program's entry point: switch the faulting MIOpen solver off unless the environment already says something
    explicit.  Returns the value in effect."""
    return os.environ.setdefault(*MIOPEN_WORKAROUND)


def _warn_if_exposed():
    name, want = MIOPEN_WORKAROUND
    if os.environ.get(name) != want:
        warnings.warn("%s is not %s: MIOpen's ConvAsmImplicitGemmGTCDynamicBwdXdlopsNHWC solver can fault in the backward of these "
                      "nets on gfx950 (INTEGRATION.md, launch requirements); call nets.coarse_net.apply_miopen_workaround() before "
                      "the first convolution runs" % (name, want), RuntimeWarning, stacklevel=3)


def _conv_bn_relu(cin, cout, k, stride=1, act=True):
    layers = [nn.Conv2d(cin, cout, k, stride=stride, padding=k // 2, bias=False), nn.BatchNorm2d(cout)]
    if act:
        layers.append(nn.ReLU(inplace=True))
    return nn.Sequential(*layers)


class Bottleneck(nn.Module):
    """slim resnet_v1 bottleneck unit: 1x1 -> 3x3 (stride) -> 1x1 (x4), post-activation, projection shortcut when the
    depth changes, sub-sampled identity otherwise."""

    def __init__(self, cin, base, stride):
        super().__init__()
        cout = base * 4
        self.conv1 = _conv_bn_relu(cin, base, 1)
        self.conv2 = _conv_bn_relu(base, base, 3, stride)
        self.conv3 = _conv_bn_relu(base, cout, 1, act=False)
        if cin != cout:
            self.shortcut = _conv_bn_relu(cin, cout, 1, stride, act=False)
        elif stride != 1:
            self.shortcut = nn.MaxPool2d(1, stride)
        else:
            self.shortcut = nn.Identity()
        self.relu = nn.ReLU(inplace=True)

    def forward(self, x):
        return self.relu(self.conv3(self.conv2(self.conv1(x))) + self.shortcut(x))


def _block(cin, base, units, stride):
    mods, c = [], cin
    for u in range(units):
        mods.append(Bottleneck(c, base, stride if u == units - 1 else 1))
        c = base * 4
    return nn.Sequential(*mods), c


class CoarseNetIter(nn.Module):
    """One CoarseNet iteration: [B,H,W,7] -> raw 235-d prediction (before set_constraints)."""

    def __init__(self, ndim=235, in_channels=7):
        super().__init__()
        self.conv_in = _conv_bn_relu(in_channels, 64, 7, 2)
        blocks, c = [], 64
        for base, stride in ((64, 1), (128, 2), (256, 2), (512, 2)):
            b, c = _block(c, base, 2, stride)
            blocks.append(b)
        self.blocks = nn.Sequential(*blocks)
        self.conv_out = _conv_bn_relu(c, ndim, 3)
        self.fc = nn.Linear(ndim, ndim)
        nn.init.normal_(self.fc.weight, mean=0.0, std=0.001)  # network.py:134
        nn.init.zeros_(self.fc.bias)

    def forward(self, net_input_nhwc):
        x = net_input_nhwc.permute(0, 3, 1, 2).contiguous()
        x = self.conv_out(self.blocks(self.conv_in(x)))
        x = x.mean(dim=(2, 3))          # global average pooling, network.py:132
        return self.fc(x)               # (B, ndim)


class CoarseNet(nn.Module):
    """nIter iterations, each with its own weights (network.py:113-136), around the decode -> render hot path."""

    def __init__(self, face_net, nIter=4):
        super().__init__()
        _warn_if_exposed()
        self.face_net = face_net        # nets.network.FaceRecNet (holds the 3DMM constants on the GPU)
        self.iters = nn.ModuleList([CoarseNetIter(face_net.ndim) for _ in range(nIter)])

    def forward(self, im_gray, pred_params=None):
        fn = self.face_net
        B = im_gray.shape[0]
        # (the reference rebinds self.pred_params while building its static graph; an eager forward starts from the
        # constant initial parameters every time)
        params = fn.init_pred_params[:B] if pred_params is None else pred_params
        params = params.reshape(B, fn.ndim).to(im_gray.device)
        for it in self.iters:
            vertices_proj = fn.vertices_transform(params)                       # Input_Rendering_iter%d, :113
            net_input, _ = fn.coarse_net_input(vertices_proj, im_gray=im_gray)  # :116-122
            params = fn.set_constraints(it(net_input)[:, None, None, :]).reshape(B, fn.ndim)
        return params

    def depth(self, im_gray, pred_params):
        """depth_rendering_layer (network.py:300-309) on the final parameters: coarse depth map [B,H,W,1]."""
        fn = self.face_net
        v = fn.vertices_transform(pred_params)
        return fn.coarse_net_input(v, im_gray=im_gray)[1]


class FineNet(nn.Module):
    """[im_gray | coarse depth] (B,H,W,2) -> refined depth map (B,H,W,1)  (network.py:311-333)."""

    def __init__(self):
        super().__init__()
        _warn_if_exposed()
        self.conv1 = nn.Sequential(_conv_bn_relu(2, 64, 3), _conv_bn_relu(64, 64, 3))
        self.conv2 = nn.Sequential(_conv_bn_relu(64, 128, 3), _conv_bn_relu(128, 128, 3))
        self.conv3 = nn.Sequential(_conv_bn_relu(128, 256, 3), _conv_bn_relu(256, 256, 3), _conv_bn_relu(256, 256, 3))
        self.pool = nn.MaxPool2d(2)

        def up(cin, cout):
            return nn.Sequential(nn.ConvTranspose2d(cin, cout, 2, stride=2, bias=False), nn.BatchNorm2d(cout),
                                 nn.ReLU(inplace=True))
        self.upconv2 = up(128, 128)
        self.upconv3 = nn.Sequential(up(256, 256), up(256, 256))
        self.head = nn.Sequential(_conv_bn_relu(64 + 128 + 256, 50, 1), _conv_bn_relu(50, 50, 1),
                                  _conv_bn_relu(50, 10, 1), nn.Conv2d(10, 1, 1))

    def forward(self, im_gray, coarse_depth):
        x = torch.cat([im_gray, coarse_depth], dim=3).permute(0, 3, 1, 2).contiguous()
        c1 = self.conv1(x)
        c2 = self.conv2(self.pool(c1))
        c3 = self.conv3(self.pool(c2))
        h = torch.cat([c1, self.upconv2(c2), self.upconv3(c3)], dim=1)
        return self.head(h).permute(0, 2, 3, 1).contiguous()


class FaceReconModel(nn.Module):
    """FaceRecNet.build() as ONE module (network.py:69-101): CoarseNet x nIter -> depth rendering layer -> FineNet.

    forward(im_gray) returns {'pred_params' (B,d), 'vertices_proj' (B,3,N), 'coarse_depth_map' (B,H,W,1),
    'pred_depth_map' (B,H,W,1) or None}.  Wrap THIS module in DistributedDataParallel and call the wrapper, so that DDP's
    forward runs (reducer.prepare_for_backward) and the RCCL all-reduce of the gradients actually happens."""

    def __init__(self, face_net, nIter=4, fine=True):
        super().__init__()
        self.face_net = face_net
        self.coarse = CoarseNet(face_net, nIter=nIter)
        self.fine = FineNet() if fine else None

    def forward(self, im_gray, with_depth=True):
        fn = self.face_net
        params = self.coarse(im_gray)
        out = {"pred_params": params, "vertices_proj": None, "coarse_depth_map": None, "pred_depth_map": None}
        if with_depth or self.fine is not None:
            v = fn.vertices_transform(params)                              # depth_rendering_layer, network.py:300-309
            out["vertices_proj"] = v
            out["coarse_depth_map"] = fn.coarse_net_input(v, im_gray=im_gray)[1]
            if self.fine is not None:
                out["pred_depth_map"] = self.fine(im_gray, out["coarse_depth_map"])
        return out
