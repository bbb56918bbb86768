"""CPU restatement of the reference's image augmentations with the random draws as ARGUMENTS (the reference draws them inside each operator).
TEST INFRASTRUCTURE: only tests/, oracle/make_golden_aug.py and bench-side checkers may import this; the product path never does.

Restates (all paths relative to /root/reference/src):
  utils/diffaug.py:47-50   rand_brightness     x + (r - 0.5)
  utils/diffaug.py:53-56   rand_saturation     (x - mean_c x) * (2 r) + mean_c x
  utils/diffaug.py:59-62   rand_contrast       (x - mean_chw x) * (r + 0.5) + mean_chw x
  utils/diffaug.py:65-76   rand_translation    out[i][j] = zero-framed x at [i + tx][j + ty]
  utils/diffaug.py:79-95   rand_cutout         x * mask, mask = 0 on the (clamped) window around (cx, cy)
  utils/cr.py:24-31        random_flip         images with coin < p mirrored along the width
  utils/cr.py:33-48        random_translation  out[i][j] = reflect-padded x at [i + tx][j + ty]
  utils/apa_aug.py:10-21   apply_apa_aug       fake where the per-image draw < p, else real
Pinned against the reference's own functions under a seeded generator by oracle/make_golden_aug.py (bit-identical; see tests/test_aug_cpu.py).
Plain torch on CPU, differentiable through autograd to any order (that is how the gradient and second-order vectors of the fixture are made)."""
import torch


def brightness(x, b):
    return x + b.reshape(-1, 1, 1, 1)


def saturation(x, s):
    m = x.mean(dim=1, keepdim=True)
    return (x - m) * s.reshape(-1, 1, 1, 1) + m


def contrast(x, c):
    m = x.mean(dim=[1, 2, 3], keepdim=True)
    return (x - m) * c.reshape(-1, 1, 1, 1) + m


def translation(x, tx, ty):
    """zero fill: rows / columns that would come from outside the image are 0 (diffaug.py pads one zero pixel and clamps the index into it)"""
    N, _, H, W = x.shape
    rows = []
    for n in range(N):
        a, b = int(tx[n]), int(ty[n])
        out = torch.zeros_like(x[n])
        i0, i1 = max(0, -a), min(H, H - a)
        j0, j1 = max(0, -b), min(W, W - b)
        if i0 < i1 and j0 < j1:
            out = torch.nn.functional.pad(x[n][:, i0 + a:i1 + a, j0 + b:j1 + b], [j0, W - j1, i0, H - i1])
        rows.append(out)
    return torch.stack(rows, 0)


def cutout_mask(shape, cx, cy, cut_h, cut_w):
    N, _, H, W = shape
    mask = torch.ones(N, H, W)
    for n in range(N):
        r = torch.clamp(torch.arange(cut_h) + int(cx[n]) - cut_h // 2, 0, H - 1)      # diffaug.py:88-91: the window's indices are clamped, not cropped
        c = torch.clamp(torch.arange(cut_w) + int(cy[n]) - cut_w // 2, 0, W - 1)
        mask[n][r[:, None], c[None, :]] = 0
    return mask


def cutout(x, cx, cy, cut_h, cut_w):
    return x * cutout_mask(x.shape, cx, cy, cut_h, cut_w).to(x.dtype).unsqueeze(1)


def _reflect(k, L):
    k = torch.where(k < 0, -k, k)
    return torch.where(k >= L, 2 * (L - 1) - k, k)


def cr_flip(x, flip):
    return torch.stack([torch.flip(x[n], [2]) if bool(flip[n]) else x[n] for n in range(x.shape[0])], 0)


def cr_translation(x, tx, ty):
    """reflect padding (no edge repeat), cr.py:45-47"""
    N, _, H, W = x.shape
    out = []
    for n in range(N):
        r = _reflect(torch.arange(H) + int(tx[n]), H)
        c = _reflect(torch.arange(W) + int(ty[n]), W)
        out.append(x[n].index_select(1, r).index_select(2, c))
    return torch.stack(out, 0)


def diffaug(x, policy, draws):
    """apply_diffaug (diffaug.py:35-45, channels first) with the draws of `draw_diffaug` (a list in consumption order)"""
    it = iter(draws)
    H, W = x.shape[2], x.shape[3]
    for p in policy.split(","):
        if p == "color":
            x = brightness(x, next(it) - 0.5)
            x = saturation(x, next(it) * 2)
            x = contrast(x, next(it) + 0.5)
        elif p == "translation":
            x = translation(x, next(it).reshape(-1), next(it).reshape(-1))
        elif p == "cutout":
            x = cutout(x, next(it).reshape(-1), next(it).reshape(-1), int(H * 0.5 + 0.5), int(W * 0.5 + 0.5))
        else:
            raise KeyError(p)
    return x.contiguous()


def draw_diffaug(shape, policy, dtype=torch.float32, device="cpu"):
    """the random numbers apply_diffaug consumes, drawn with the reference's own calls in its order (diffaug.py:48,54,60,66-67,81-82)"""
    N, _, H, W = shape
    out = []
    for p in policy.split(","):
        if p == "color":
            out += [torch.rand(N, 1, 1, 1, dtype=dtype, device=device) for _ in range(3)]
        elif p == "translation":
            sx, sy = int(H * 0.125 + 0.5), int(W * 0.125 + 0.5)
            out += [torch.randint(-sx, sx + 1, size=[N, 1, 1], device=device), torch.randint(-sy, sy + 1, size=[N, 1, 1], device=device)]
        elif p == "cutout":
            ch, cw = int(H * 0.5 + 0.5), int(W * 0.5 + 0.5)
            out += [torch.randint(0, H + (1 - ch % 2), size=[N, 1, 1], device=device), torch.randint(0, W + (1 - cw % 2), size=[N, 1, 1], device=device)]
    return out


def draw_cr(shape, flip=True, translation=True, device="cpu"):
    """cr.py:27,35-36: the coin comes from the CPU generator, the shifts from the images' device"""
    N, _, H, W = shape
    coin = torch.FloatTensor(N, 1).uniform_(0.0, 1.0) if flip else None
    tx = ty = None
    if translation:
        mx, my = int(H * (1 / 8)), int(W * (1 / 8))
        tx = torch.randint(-mx, mx + 1, size=[N, 1, 1], device=device)
        ty = torch.randint(-my, my + 1, size=[N, 1, 1], device=device)
    return coin, tx, ty


def cr_aug(x, coin, tx, ty):
    if coin is not None:
        x = cr_flip(x, (coin < 0.5).reshape(-1))
    if tx is not None:
        x = cr_translation(x, tx.reshape(-1), ty.reshape(-1))
    return x.contiguous()


def apa(real, fake, coin, p):
    """utils/apa_aug.py:10-21: the real image of a slot is replaced by the fake one where the uniform draw is below p"""
    return torch.where((coin < p).reshape(-1, 1, 1, 1), fake, real)


def mse(a, b):
    return ((a - b) ** 2).mean()
