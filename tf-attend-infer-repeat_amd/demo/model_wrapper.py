"""Inference wrapper -- counterpart of the reference's demo/model_wrapper.py:4-52.

`ModelWrapper(model).infer(images)` takes a list of canvases and returns the same six per-image
lists (digit counts, [s, x, y] positions, reconstructions, attention windows, latents, losses).
The reference feeds a placeholder through a session; here the images are copied into the model's
resident input buffer (its batch size is fixed at construction), in chunks, padded with blank
canvases, and the outputs are read back once per chunk.
"""
import numpy as np
import torch


class ModelWrapper:

    def __init__(self, model, session=None, data_placeholder=None, canvas_size=50, window_size=28):
        self.model = model
        self.session = session                      # accepted for call-site parity, unused
        self.data_placeholder = data_placeholder    # ditto
        self.canvas_size = canvas_size
        self.window_size = window_size

    def infer(self, images):
        """Returns (digit counts, [n_i, 3] (s, x, y) rows, [C, C] reconstructions, [n_i, w, w] windows,
        [n_i, Z] latents, reconstruction losses) -- one entry per image, n_i = inferred count."""
        m, B, cs, ws = self.model, self.model.batch_size, self.canvas_size, self.window_size
        flat = np.zeros((len(images), cs * cs), np.float32)
        for i, img in enumerate(images):
            flat[i] = np.asarray(img, dtype=np.float32).reshape(-1)
        out = ([], [], [], [], [], [])
        staging = torch.zeros(B, cs * cs)
        for lo in range(0, len(flat), B):
            k = min(B, len(flat) - lo)
            staging.zero_()                                   # blank canvases pad the last chunk
            staging[:k] = torch.from_numpy(flat[lo:lo + k])
            m.input_images.copy_(staging)
            m.forward()
            # the fetch set of the reference wrapper (model_wrapper.py:19-25), sliced to this chunk
            counts = m.rec_num_digits[:k].cpu().numpy().astype(int)
            sxy = torch.cat([m.rec_scales[:k], m.rec_shifts[:k]], dim=2).cpu().numpy()          # [k, T', 3]
            wins = m.rec_windows[:k].cpu().numpy().reshape(k, -1, ws, ws)
            lats = m.rec_latents[:k].cpu().numpy()
            recs = m.reconstruction[:k].cpu().numpy().reshape(k, cs, cs)
            loss = m.reconstruction_loss[:k].cpu().numpy()
            # only the first `count` steps of an image are objects; an image without objects gets the
            # empty array the reference's np.array([]) produces (shape (0,))
            none = np.array([])
            out[0].extend(int(c) for c in counts)
            out[1].extend(sxy[i, :c] if c else none for i, c in enumerate(counts))
            out[2].extend(recs)
            out[3].extend(wins[i, :c] if c else none for i, c in enumerate(counts))
            out[4].extend(lats[i, :c] if c else none for i, c in enumerate(counts))
            out[5].extend(loss)
        return out
