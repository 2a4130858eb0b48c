"""Attention-window visualisation -- the image summary of the reference
(air_model.py:130-157 `_draw_colored_bounding_boxes`, :211-267 `_visualize_reconstructions`,
:634-647) as device tensors: original and reconstruction enlarged `zoom` times, the attention
windows of up to three steps rasterised through the spatial transformer (the HIP kernel behind
air/transformer.py) from the backward ST matrices and drawn in R, G, B.
"""
import torch

from .transformer import transformer


def resize_bilinear_tf1(images, out_h, out_w):
    """tf.image.resize_images(...) of TF 1.3 = resize_bilinear(align_corners=False): source
    coordinate = destination * (in / out), no half-pixel offset, upper neighbour clamped.
    images [B, H, W] -> [B, out_h, out_w]."""
    B, H, W = images.shape
    dev = images.device
    ys = torch.arange(out_h, device=dev, dtype=torch.float32) * (H / out_h)
    xs = torch.arange(out_w, device=dev, dtype=torch.float32) * (W / out_w)
    y0, x0 = ys.floor().long(), xs.floor().long()
    y1, x1 = (y0 + 1).clamp(max=H - 1), (x0 + 1).clamp(max=W - 1)
    wy, wx = (ys - y0.float()).view(1, -1, 1), (xs - x0.float()).view(1, 1, -1)
    top = images[:, y0][:, :, x0] * (1 - wx) + images[:, y0][:, :, x1] * wx
    bot = images[:, y1][:, :, x0] * (1 - wx) + images[:, y1][:, :, x1] * wx
    return top * (1 - wy) + bot * wy


def draw_colored_bounding_boxes(images, boxes, steps):
    """air_model.py:130-157.  images [B, H, W], boxes [B, S>=3, H, W] in {0,1}, steps [B] ->
    [B, H, W, 3]: the s-th box is added to channel s and subtracted from the other two, for the
    images that took more than s steps."""
    channels = [images, images, images]
    for s in range(min(3, boxes.shape[1])):
        box = boxes[:, s]
        on = (steps > s).view(-1, 1, 1)
        for c in range(3):
            if s == c:
                channels[c] = torch.where(on, torch.minimum(channels[c] + box, torch.ones_like(images)), channels[c])
            else:
                channels[c] = torch.where(on, torch.maximum(channels[c] - box, torch.zeros_like(images)), channels[c])
    return torch.stack(channels, dim=3)


def visualize_reconstructions(original, reconstruction, st_back, steps, canvas_size, windows_size, max_steps, zoom=2):
    """air_model.py:211-267.  original / reconstruction [n, C*C], st_back [n, T', 2, 3], steps [n]
    -> [n, zoom*C, 2*zoom*C + 4, 3] float in [0, 1]: original with boxes | white stripe | reconstruction
    with boxes."""
    n, C, w, Z = original.shape[0], canvas_size, windows_size, zoom * canvas_size
    large_o = resize_bilinear_tf1(original.reshape(n, C, C).float(), Z, Z)
    large_r = resize_bilinear_tf1(reconstruction.reshape(n, C, C).float(), Z, Z)
    # pad the ST matrices to max_steps (fewer steps may have been executed globally, :226-231)
    T = st_back.shape[1]
    if T < max_steps:
        st_back = torch.cat([st_back, torch.zeros(n, max_steps - T, 2, 3, device=st_back.device)], dim=1)
    # a window-sized frame with a one-pixel border (tf.image.draw_bounding_boxes with box [0,0,1,1])
    frame = torch.zeros(w, w, device=original.device)
    frame[0, :] = 1.0; frame[-1, :] = 1.0; frame[:, 0] = 1.0; frame[:, -1] = 1.0
    frames = frame.expand(n * max_steps, w, w).contiguous()
    boxes = transformer(frames.unsqueeze(3), st_back.reshape(n * max_steps, 6).contiguous(), [Z, Z])
    boxes = boxes.reshape(n, max_steps, Z, Z).clamp(0.0, 1.0)
    boxes = (boxes > 0.01).float()                                       # sharpen the borders :250-254
    left = draw_colored_bounding_boxes(large_o, boxes, steps)
    right = draw_colored_bounding_boxes(large_r, boxes, steps)
    stripe = torch.ones(n, Z, 4, 3, device=original.device)
    return torch.cat([left, stripe, right], dim=2)


def save_image_grid(images, path, columns=10, pad=2):
    """[n, H, W, 3] float in [0,1] -> one PNG (rows of `columns` images), like images/rec_samples.png."""
    import numpy as np
    from PIL import Image
    a = (images.detach().clamp(0, 1) * 255.0).round().to(torch.uint8).cpu().numpy()
    n, H, W, _ = a.shape
    rows = (n + columns - 1) // columns
    grid = np.full((rows * (H + pad) + pad, columns * (W + pad) + pad, 3), 255, np.uint8)
    for i in range(n):
        r, c = divmod(i, columns)
        grid[pad + r * (H + pad): pad + r * (H + pad) + H, pad + c * (W + pad): pad + c * (W + pad) + W] = a[i]
    Image.fromarray(grid).save(path)
    return path
