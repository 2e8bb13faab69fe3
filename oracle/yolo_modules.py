"""TEST INFRASTRUCTURE — second, structurally independent restatement of YOLOv5s (never imported by effocr_amd/).

oracle/yolo_ref.py writes the network out as a flat sequence of ``torch.nn.functional`` calls.  This file instead
re-creates it the way the ultralytics repository itself does: the published ``models/yolov5s.yaml`` layer table
(``[from, number, module, args]`` rows, depth_multiple 0.33, width_multiple 0.50, the three anchor rows) is kept as DATA
and turned into ``torch.nn`` modules by a small ``parse_model`` (channel bookkeeping with make_divisible(c * width, 8),
repeat counts max(round(n * depth), 1), ``from`` indices resolved against the list of saved outputs).  Parameter names
fall out of the module tree (``model.<layer>.conv.weight``, ``model.<layer>.m.<i>.cv1.bn.running_mean``,
``model.24.m.<level>.bias``, ``model.24.anchors``) instead of being typed in by hand.

The two restatements share no code; tests/test_oracle.py checks that
  * the module tree's state-dict keys and shapes are exactly those the product's localizer expects,
  * it has the PUBLISHED parameter count of YOLOv5s v6.x at 80 classes (7,235,389) and conv FLOPs (~16.5 GFLOPs at 640),
  * a forward pass with a seeded state dict equals oracle/yolo_ref.py's to fp32 round-off.
Still PARITY UNPINNED in the strict sense (the reference ships an ONNX export, not weights or a definition; ultralytics
is not installable here): this is a consistency anchor between two readings of the published architecture.
(reference call sites: onnx_engines/localizer_engine.py:24-28,49-66)
"""
import math

import torch
import torch.nn as nn

# models/yolov5s.yaml (v6.x) as data
DEPTH_MULTIPLE, WIDTH_MULTIPLE = 0.33, 0.50
ANCHORS = ((10, 13, 16, 30, 33, 23), (30, 61, 62, 45, 59, 119), (116, 90, 156, 198, 373, 326))
BACKBONE = (
    (-1, 1, "Conv", (64, 6, 2, 2)),       # 0-P1/2
    (-1, 1, "Conv", (128, 3, 2)),         # 1-P2/4
    (-1, 3, "C3", (128,)),
    (-1, 1, "Conv", (256, 3, 2)),         # 3-P3/8
    (-1, 6, "C3", (256,)),
    (-1, 1, "Conv", (512, 3, 2)),         # 5-P4/16
    (-1, 9, "C3", (512,)),
    (-1, 1, "Conv", (1024, 3, 2)),        # 7-P5/32
    (-1, 3, "C3", (1024,)),
    (-1, 1, "SPPF", (1024, 5)),           # 9
)
HEAD = (
    (-1, 1, "Conv", (512, 1, 1)),
    (-1, 1, "Upsample", (None, 2, "nearest")),
    ((-1, 6), 1, "Concat", (1,)),         # cat backbone P4
    (-1, 3, "C3", (512, False)),          # 13
    (-1, 1, "Conv", (256, 1, 1)),
    (-1, 1, "Upsample", (None, 2, "nearest")),
    ((-1, 4), 1, "Concat", (1,)),         # cat backbone P3
    (-1, 3, "C3", (256, False)),          # 17 (P3/8-small)
    (-1, 1, "Conv", (256, 3, 2)),
    ((-1, 14), 1, "Concat", (1,)),        # cat head P4
    (-1, 3, "C3", (512, False)),          # 20 (P4/16-medium)
    (-1, 1, "Conv", (512, 3, 2)),
    ((-1, 10), 1, "Concat", (1,)),        # cat head P5
    (-1, 3, "C3", (1024, False)),         # 23 (P5/32-large)
    ((17, 20, 23), 1, "Detect", ()),      # Detect(P3, P4, P5)
)


def make_divisible(x, divisor):
    return math.ceil(x / divisor) * divisor


def autopad(k, p=None):
    return k // 2 if p is None else p


class Conv(nn.Module):
    def __init__(self, c1, c2, k=1, s=1, p=None):
        super().__init__()
        self.conv = nn.Conv2d(c1, c2, k, s, autopad(k, p), bias=False)
        self.bn = nn.BatchNorm2d(c2, eps=1e-3, momentum=0.03)
        self.act = nn.SiLU()

    def forward(self, x):
        return self.act(self.bn(self.conv(x)))


class Bottleneck(nn.Module):
    def __init__(self, c1, c2, shortcut=True, e=0.5):
        super().__init__()
        c_ = int(c2 * e)
        self.cv1 = Conv(c1, c_, 1, 1)
        self.cv2 = Conv(c_, c2, 3, 1)
        self.add = shortcut and c1 == c2

    def forward(self, x):
        return x + self.cv2(self.cv1(x)) if self.add else self.cv2(self.cv1(x))


class C3(nn.Module):
    def __init__(self, c1, c2, n=1, shortcut=True, e=0.5):
        super().__init__()
        c_ = int(c2 * e)
        self.cv1 = Conv(c1, c_, 1, 1)
        self.cv2 = Conv(c1, c_, 1, 1)
        self.cv3 = Conv(2 * c_, c2, 1)
        self.m = nn.Sequential(*(Bottleneck(c_, c_, shortcut, e=1.0) for _ in range(n)))

    def forward(self, x):
        return self.cv3(torch.cat((self.m(self.cv1(x)), self.cv2(x)), 1))


class SPPF(nn.Module):
    def __init__(self, c1, c2, k=5):
        super().__init__()
        c_ = c1 // 2
        self.cv1 = Conv(c1, c_, 1, 1)
        self.cv2 = Conv(c_ * 4, c2, 1, 1)
        self.m = nn.MaxPool2d(kernel_size=k, stride=1, padding=k // 2)

    def forward(self, x):
        x = self.cv1(x)
        y1 = self.m(x)
        y2 = self.m(y1)
        return self.cv2(torch.cat((x, y1, y2, self.m(y2)), 1))


class Concat(nn.Module):
    def __init__(self, dimension=1):
        super().__init__()
        self.d = dimension

    def forward(self, xs):
        return torch.cat(xs, self.d)


class Detect(nn.Module):
    """models/yolo.py Detect, inference branch (the tensor the exported model returns first)."""

    def __init__(self, nc, anchors, ch, strides=(8.0, 16.0, 32.0)):
        super().__init__()
        self.nc, self.no, self.nl, self.na = nc, nc + 5, len(anchors), len(anchors[0]) // 2
        self.stride = torch.tensor(strides)
        a = torch.tensor(anchors, dtype=torch.float32).view(self.nl, -1, 2)
        self.register_buffer("anchors", a / self.stride.view(-1, 1, 1))      # stride units, as in the checkpoint
        self.m = nn.ModuleList(nn.Conv2d(c, self.no * self.na, 1) for c in ch)

    def forward(self, feats):
        z = []
        for i, f in enumerate(feats):
            r = self.m[i](f)
            bs, _, ny, nx = r.shape
            y = r.view(bs, self.na, self.no, ny, nx).permute(0, 1, 3, 4, 2).contiguous().sigmoid()
            yv, xv = torch.meshgrid(torch.arange(ny, dtype=r.dtype), torch.arange(nx, dtype=r.dtype), indexing="ij")
            grid = torch.stack((xv, yv), 2).expand(1, self.na, ny, nx, 2) - 0.5
            anchor_grid = (self.anchors[i] * self.stride[i]).view(1, self.na, 1, 1, 2).expand(1, self.na, ny, nx, 2)
            xy = (y[..., 0:2] * 2 + grid) * self.stride[i]
            wh = (y[..., 2:4] * 2) ** 2 * anchor_grid
            z.append(torch.cat((xy, wh, y[..., 4:]), -1).view(bs, -1, self.no))
        return torch.cat(z, 1)


class YoloV5(nn.Module):
    """parse_model over the yaml rows above: ``self.model`` is the nn.Sequential whose indices are the layer numbers."""

    def __init__(self, nc=80, depth=DEPTH_MULTIPLE, width=WIDTH_MULTIPLE):
        super().__init__()
        ch, layers, self.froms = [3], [], []
        for f, n, kind, args in BACKBONE + HEAD:
            n = max(round(n * depth), 1) if n > 1 else n
            src = ch[f] if isinstance(f, int) else None
            if kind in ("Conv", "C3", "SPPF"):
                c2 = make_divisible(args[0] * width, 8)
                if kind == "Conv":
                    m = Conv(src, c2, *args[1:])
                elif kind == "C3":
                    m = C3(src, c2, n, *args[1:])
                else:
                    m = SPPF(src, c2, *args[1:])
            elif kind == "Upsample":
                m, c2 = nn.Upsample(None, args[1], args[2]), src
            elif kind == "Concat":
                m, c2 = Concat(*args), sum(ch[j] for j in f)
            elif kind == "Detect":
                m, c2 = Detect(nc, ANCHORS, [ch[j] for j in f]), None
            else:
                raise ValueError(kind)
            layers.append(m)
            self.froms.append(f)
            if len(layers) == 1:
                ch = []                                  # from here on ch[j] = output channels of layer j (ultralytics parse_model)
            ch.append(c2)
        self.model = nn.Sequential(*layers)

    def forward(self, x):
        saved = []
        for m, f in zip(self.model, self.froms):
            if not isinstance(f, int):
                x = [x if j == -1 else saved[j] for j in f]
            elif f != -1:
                x = saved[f]
            x = m(x)
            saved.append(x)
        return x


def conv_flops(model, h, w):
    """2 x multiply-accumulates of every Conv2d for one h x w image (hooks; what thop / ultralytics' model summary counts)."""
    total = [0.0]

    def hook(mod, inp, out):
        k = mod.kernel_size[0] * mod.kernel_size[1] * mod.in_channels // mod.groups
        total[0] += 2.0 * k * out.numel() / out.shape[0]

    hs = [m.register_forward_hook(hook) for m in model.modules() if isinstance(m, nn.Conv2d)]
    with torch.no_grad():
        model.eval()(torch.zeros(1, 3, h, w))
    for hdl in hs:
        hdl.remove()
    return total[0]
