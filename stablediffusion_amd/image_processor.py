"""Host-side image / mask pre-processing with the call surface of diffusers' `VaeImageProcessor`, as the
reference holds it on `SDModelWrapper` (`/root/reference/models/stable_diffusion.py:96-101`) and calls it at
`/root/reference/pipelines/sd_unified_pipeline.py:238` (img2img), `:271` (get_crop_region), `:278-285`
(inpaint image and mask).  PIL / numpy / torch inputs -> float32 NCHW tensors (images in [-1, 1], masks
binarised to {0, 1}, one channel).

diffusers 0.27.2 is not installed here, so this is a restatement of its published behaviour (resize to a
multiple of the VAE scale factor with Lanczos resampling, /255, 2x-1 normalisation, 0.5 binarisation,
`convert("L")` for masks, crop region grown to the target aspect ratio, "fill" resize) -- PARITY UNPINNED,
like the oracle.  Pure host code (PIL + numpy + torch); nothing here touches the GPU.
"""
from __future__ import annotations

from typing import List, Optional, Tuple, Union

import numpy as np
import torch

try:                                   # PIL is importable in this image; keep the tensor path usable without it
    from PIL import Image
except ImportError:                    # pragma: no cover
    Image = None

_RESAMPLE = {"lanczos": "LANCZOS", "bilinear": "BILINEAR", "bicubic": "BICUBIC", "nearest": "NEAREST"}


class VaeImageProcessor:
    def __init__(self, do_resize: bool = True, vae_scale_factor: int = 8, resample: str = "lanczos",
                 do_normalize: bool = True, do_binarize: bool = False, do_convert_rgb: bool = False,
                 do_convert_grayscale: bool = False):
        if do_convert_rgb and do_convert_grayscale:
            raise ValueError("`do_convert_rgb` and `do_convert_grayscale` can not both be set to `True`")
        self.do_resize, self.vae_scale_factor, self.resample = do_resize, int(vae_scale_factor), resample
        self.do_normalize, self.do_binarize = do_normalize, do_binarize
        self.do_convert_rgb, self.do_convert_grayscale = do_convert_rgb, do_convert_grayscale

    # ---- conversions -----------------------------------------------------------------------------
    @staticmethod
    def pil_to_numpy(images) -> np.ndarray:
        if not isinstance(images, list):
            images = [images]
        return np.stack([np.array(im).astype(np.float32) / 255.0 for im in images], axis=0)

    @staticmethod
    def numpy_to_pil(images: np.ndarray):
        if images.ndim == 3:
            images = images[None]
        images = (images * 255).round().astype("uint8")
        if images.shape[-1] == 1:
            return [Image.fromarray(im.squeeze(), mode="L") for im in images]
        return [Image.fromarray(im) for im in images]

    @staticmethod
    def numpy_to_pt(images: np.ndarray) -> torch.Tensor:
        if images.ndim == 3:
            images = images[..., None]
        return torch.from_numpy(images.transpose(0, 3, 1, 2))

    @staticmethod
    def pt_to_numpy(images: torch.Tensor) -> np.ndarray:
        return images.cpu().permute(0, 2, 3, 1).float().numpy()

    @staticmethod
    def normalize(images):
        return 2.0 * images - 1.0

    @staticmethod
    def denormalize(images):
        return (images / 2 + 0.5).clamp(0, 1)

    @staticmethod
    def binarize(image):
        image[image < 0.5] = 0
        image[image >= 0.5] = 1
        return image

    # ---- geometry --------------------------------------------------------------------------------
    def get_default_height_width(self, image, height: Optional[int] = None, width: Optional[int] = None) -> Tuple[int, int]:
        if height is None:
            height = image.height if Image is not None and isinstance(image, Image.Image) else (
                image.shape[2] if isinstance(image, torch.Tensor) else image.shape[1])
        if width is None:
            width = image.width if Image is not None and isinstance(image, Image.Image) else (
                image.shape[3] if isinstance(image, torch.Tensor) else image.shape[2])
        f = self.vae_scale_factor
        return height - height % f, width - width % f

    def _pil_resample(self):
        return getattr(Image.Resampling, _RESAMPLE[self.resample])

    def _resize_and_fill(self, image, width: int, height: int):
        """Fit inside (width, height) keeping the aspect ratio, centre it, fill the bars by stretching the edge."""
        ratio, src_ratio = width / height, image.width / image.height
        src_w = width if ratio < src_ratio else image.width * height // image.height
        src_h = height if ratio >= src_ratio else image.height * width // image.width
        resized = image.resize((src_w, src_h), resample=self._pil_resample())
        res = Image.new("RGB", (width, height))
        res.paste(resized, box=(width // 2 - src_w // 2, height // 2 - src_h // 2))
        if ratio < src_ratio:
            fill = height // 2 - src_h // 2
            if fill > 0:
                res.paste(resized.resize((width, fill), box=(0, 0, width, 0)), box=(0, 0))
                res.paste(resized.resize((width, fill), box=(0, resized.height, width, resized.height)), box=(0, fill + src_h))
        elif ratio > src_ratio:
            fill = width // 2 - src_w // 2
            if fill > 0:
                res.paste(resized.resize((fill, height), box=(0, 0, 0, height)), box=(0, 0))
                res.paste(resized.resize((fill, height), box=(resized.width, 0, resized.width, height)), box=(fill + src_w, 0))
        return res

    def _resize_and_crop(self, image, width: int, height: int):
        ratio, src_ratio = width / height, image.width / image.height
        src_w = width if ratio > src_ratio else image.width * height // image.height
        src_h = height if ratio <= src_ratio else image.height * width // image.width
        resized = image.resize((src_w, src_h), resample=self._pil_resample())
        res = Image.new("RGB", (width, height))
        res.paste(resized, box=(width // 2 - src_w // 2, height // 2 - src_h // 2))
        return res

    def resize(self, image, height: int, width: int, resize_mode: str = "default"):
        if resize_mode != "default" and not (Image is not None and isinstance(image, Image.Image)):
            raise ValueError(f"Only PIL image input is supported for resize_mode {resize_mode}")
        if Image is not None and isinstance(image, Image.Image):
            if resize_mode == "default":
                return image.resize((width, height), resample=self._pil_resample())
            if resize_mode == "fill":
                return self._resize_and_fill(image, width, height)
            if resize_mode == "crop":
                return self._resize_and_crop(image, width, height)
            raise ValueError(f"resize_mode {resize_mode} is not supported")
        if isinstance(image, torch.Tensor):
            return torch.nn.functional.interpolate(image, size=(height, width))
        t = torch.nn.functional.interpolate(self.numpy_to_pt(image), size=(height, width))
        return self.pt_to_numpy(t)

    @staticmethod
    def get_crop_region(mask_image, width: int, height: int, pad: int = 0):
        """Bounding box of the mask's non-zero area, padded by `pad`, then grown to the aspect ratio
        width : height inside the image (`sd_unified_pipeline.py:271`).  Returns (x1, y1, x2, y2)."""
        mask = np.array(mask_image.convert("L"))
        h, w = mask.shape
        cols, rows = np.nonzero(mask.any(axis=0))[0], np.nonzero(mask.any(axis=1))[0]
        if len(cols) == 0:
            x1, y1, x2, y2 = w, h, 0, 0                     # empty mask: diffusers' counters run through
        else:
            x1, x2, y1, y2 = int(cols[0]), int(cols[-1]) + 1, int(rows[0]), int(rows[-1]) + 1
        x1, y1 = max(x1 - pad, 0), max(y1 - pad, 0)
        x2, y2 = min(x2 + pad, w), min(y2 + pad, h)
        ratio_crop = (x2 - x1) / (y2 - y1) if y2 > y1 else 1.0
        ratio_proc = width / height
        if ratio_crop > ratio_proc:
            desired = (x2 - x1) / ratio_proc
            diff = int(desired - (y2 - y1))
            y1 -= diff // 2
            y2 += diff - diff // 2
            if y2 >= h:
                y1, y2 = y1 - (y2 - h), h
            if y1 < 0:
                y1, y2 = 0, y2 - y1
            y2 = min(y2, h)
        else:
            desired = (y2 - y1) * ratio_proc
            diff = int(desired - (x2 - x1))
            x1 -= diff // 2
            x2 += diff - diff // 2
            if x2 >= w:
                x1, x2 = x1 - (x2 - w), w
            if x1 < 0:
                x1, x2 = 0, x2 - x1
            x2 = min(x2, w)
        return x1, y1, x2, y2

    # ---- the two entry points the pipeline uses --------------------------------------------------
    def preprocess(self, image, height: Optional[int] = None, width: Optional[int] = None,
                   resize_mode: str = "default", crops_coords: Optional[Tuple[int, int, int, int]] = None) -> torch.Tensor:
        pil = Image is not None
        if self.do_convert_grayscale and isinstance(image, (torch.Tensor, np.ndarray)) and image.ndim == 3:
            image = image.unsqueeze(1) if isinstance(image, torch.Tensor) else np.expand_dims(image, axis=-1)
        if pil and isinstance(image, Image.Image) or isinstance(image, (np.ndarray, torch.Tensor)):
            image = [image]
        if not isinstance(image, list) or not image:
            raise ValueError("Input is in incorrect format: PIL image, numpy array, torch tensor or a list of one kind")
        first = image[0]
        if pil and isinstance(first, Image.Image):
            if crops_coords is not None:
                image = [im.crop(crops_coords) for im in image]
            if self.do_resize:
                height, width = self.get_default_height_width(image[0], height, width)
                image = [self.resize(im, height, width, resize_mode=resize_mode) for im in image]
            if self.do_convert_rgb:
                image = [im.convert("RGB") for im in image]
            elif self.do_convert_grayscale:
                image = [im.convert("L") for im in image]
            t = self.numpy_to_pt(self.pil_to_numpy(image))
        elif isinstance(first, np.ndarray):
            arr = np.concatenate(image, axis=0) if first.ndim == 4 else np.stack(image, axis=0)
            t = self.numpy_to_pt(arr)
            height, width = self.get_default_height_width(t, height, width)
            if self.do_resize:
                t = self.resize(t, height, width)
        elif isinstance(first, torch.Tensor):
            t = torch.cat(image, dim=0) if first.ndim == 4 else torch.stack(image, dim=0)
            if self.do_convert_grayscale and t.ndim == 3:
                t = t.unsqueeze(1)
            if t.shape[1] == 4:                       # latents pass through untouched
                return t
            height, width = self.get_default_height_width(t, height, width)
            if self.do_resize:
                t = self.resize(t, height, width)
        else:
            raise ValueError(f"Input is in incorrect format: {type(first)}")
        t = t.to(torch.float32)
        do_norm = self.do_normalize
        if do_norm and t.min() < 0:
            do_norm = False                           # already in [-1, 1] (diffusers warns and skips)
        if do_norm:
            t = self.normalize(t)
        if self.do_binarize:
            t = self.binarize(t)
        return t

    def postprocess(self, image: torch.Tensor, output_type: str = "pil", do_denormalize: Optional[List[bool]] = None):
        if output_type == "latent":
            return image
        flags = [self.do_normalize] * image.shape[0] if do_denormalize is None else do_denormalize
        image = torch.stack([self.denormalize(image[i]) if flags[i] else image[i] for i in range(image.shape[0])])
        if output_type == "pt":
            return image
        arr = self.pt_to_numpy(image)
        return arr if output_type == "np" else self.numpy_to_pil(arr)

    def apply_overlay(self, mask, init_image, image, crop_coords=None):
        """Paste the generated crop back over the original wherever the mask is set (inpaint, padding_mask_crop)."""
        width, height = image.width, image.height
        init_image = self.resize(init_image, width=width, height=height)
        mask = self.resize(mask, width=width, height=height)
        masked = Image.new("RGBa", (width, height))
        masked.paste(init_image.convert("RGBA").convert("RGBa"), mask=Image.eval(mask.convert("L"), lambda v: 255 - v))
        init_masked = masked.convert("RGBA")
        if crop_coords is not None:
            x, y, x2, y2 = crop_coords
            base = Image.new("RGBA", (width, height))
            base.paste(self.resize(image, height=y2 - y, width=x2 - x, resize_mode="crop"), (x, y))
            image = base.convert("RGB")
        image = image.convert("RGBA")
        image.alpha_composite(init_masked)
        return image.convert("RGB")
