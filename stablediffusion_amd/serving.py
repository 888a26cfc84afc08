"""The tail of the RunPod handler's `inference_mode` (`/root/reference/runpod-worker/handler_logic.py:170-192`):
pipeline call -> uint8 HWC arrays (`convert_pt_to_numpy`, `:21-29`) -> JPEG -> base64 strings -> {"images": [...]}.
RunPod itself (job polling, S3, W&B) is out of scope; this is the part that touches the
engine's output.  Host code: PIL encodes the JPEG exactly as the reference does (`Image.save(format="JPEG")`,
default quality)."""
from __future__ import annotations

import base64
import io
from typing import Dict, List

import numpy as np
import torch

from .pipeline import StableDiffusionUnifiedPipeline, convert_pt_to_numpy


def images_to_base64_jpeg(images) -> List[str]:
    """`handler_logic.py:180-187`: each HWC uint8 array -> contiguous -> PIL -> JPEG bytes -> base64 text."""
    from PIL import Image
    out = []
    for img in images:
        pil_img = Image.fromarray(np.ascontiguousarray(img))
        buf = io.BytesIO()
        pil_img.save(buf, format="JPEG")
        out.append(base64.b64encode(buf.getvalue()).decode("utf-8"))
    return out


def inference_mode(model, inference_config: Dict, device: str = "cuda") -> Dict[str, List[str]]:
    """`handler_logic.py:150-192` with the engine behind `model.base` / `model.vae`."""
    pipeline = StableDiffusionUnifiedPipeline(do_cfg=True, device=device)
    images = pipeline(model, **inference_config)
    if isinstance(images, torch.Tensor):
        images = convert_pt_to_numpy(images)
    return {"images": images_to_base64_jpeg(images)}
