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
        m = self.model
        B = m.batch_size
        flat = np.stack([np.ravel(np.asarray(img, dtype=np.float32)) for img in images]) if len(images) else \
            np.zeros((0, self.canvas_size ** 2), np.float32)
        all_digits, all_positions = [], []
        all_windows, all_latents = [], []
        all_reconstructions, all_loss = [], []
        for start in range(0, len(flat), B):
            chunk = flat[start:start + B]
            buf = np.zeros((B, flat.shape[1]), np.float32)
            buf[:len(chunk)] = chunk
            m.input_images.copy_(torch.from_numpy(buf))
            m.forward()
            # the fetch set of model_wrapper.py:19-25, one device->host copy each
            rec_digits = m.rec_num_digits.cpu().numpy()
            rec_scales = m.rec_scales.cpu().numpy()
            rec_shifts = m.rec_shifts.cpu().numpy()
            reconstructions = m.reconstruction.cpu().numpy()
            rec_windows = m.rec_windows.cpu().numpy()
            rec_latents = m.rec_latents.cpu().numpy()
            rec_loss = m.reconstruction_loss.cpu().numpy()
            for i in range(len(chunk)):
                digits = int(rec_digits[i])
                reconstruction = np.reshape(reconstructions[i], (self.canvas_size, self.canvas_size))
                positions, windows, latents = [], [], []
                for j in range(digits):
                    positions.append(np.array([rec_scales[i][j][0]] + list(rec_shifts[i][j])))
                    windows.append(np.reshape(rec_windows[i][j], (self.window_size, self.window_size)))
                    latents.append(rec_latents[i][j])
                all_digits.append(digits)
                all_positions.append(np.array(positions))
                all_reconstructions.append(reconstruction)
                all_windows.append(np.array(windows))
                all_latents.append(np.array(latents))
                all_loss.append(rec_loss[i])
        return all_digits, all_positions, all_reconstructions, all_windows, all_latents, all_loss
