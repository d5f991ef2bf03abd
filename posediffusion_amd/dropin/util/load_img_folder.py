"""load_and_preprocess_images (pose_diffusion/util/load_img_folder.py:15-48) with the crop + resize on the device.

Decoding stays on the host (PIL, as in the reference, :58-63); each frame is uploaded as uint8 HWC and one kernel
(pd_preprocess_image) does /255, the centre crop to a square (:68-73) and torch's align_corners=False bilinear resize
(:35-40).  Returns the reference's (images_tensor [N,3,S,S] float32 -- here resident on the GPU --, image_info)."""
import os

import numpy as np
import torch

from posediffusion_amd import _lib


def _bbox_and_scale(h: int, w: int, image_size: int):
    """:68-124 for box_crop_context = 0: the crop as xyxy (clamped to the image, rounded), and image_size / min(h, w)."""
    min_dim = min(h, w)
    if min_dim <= 1:
        raise ValueError("squashed image!! The bounding box contains no pixels.")          # :95-96
    top, left = (h - min_dim) // 2, (w - min_dim) // 2
    x0, y0 = min(max(left, 0), w), min(max(top, 0), h)
    x1, y1 = min(max(left + max(min_dim, 2), 0), w), min(max(top + max(min_dim, 2), 0), h)
    return np.array([x0, y0, x1, y1], dtype=np.int64), min_dim, image_size / min_dim


def load_and_preprocess_images(folder_path=None, image_size: int = 224, image_paths=None, mode: str = "bilinear", device=None):
    from PIL import Image
    if mode != "bilinear":
        raise NotImplementedError("the HIP preprocessing kernel implements mode='bilinear' (the reference's default)")
    if image_paths is None:
        image_paths = [os.path.join(folder_path, f) for f in os.listdir(folder_path) if f.lower().endswith((".png", ".jpg", ".jpeg"))]
    image_paths.sort()                                                                      # :22
    if not torch.cuda.is_available():
        raise RuntimeError("posediffusion_amd preprocessing runs only on an AMD GPU")
    dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
    lib = _lib.load()
    out = torch.empty(len(image_paths), 3, image_size, image_size, device=dev, dtype=torch.float32)
    bboxes, scales, min_hw = [], [], None
    with torch.cuda.device(dev):                       # the launches below target `dev`'s current stream
        stream = torch.cuda.current_stream(dev).cuda_stream
        for k, path in enumerate(image_paths):
            with Image.open(path) as pil_im:
                im = np.ascontiguousarray(np.array(pil_im.convert("RGB")))                  # :58-60, uint8 HWC
            h, w = im.shape[:2]
            bbox, min_hw, scale = _bbox_and_scale(h, w, image_size)
            src = torch.from_numpy(im).to(dev)
            _lib.check(lib.pd_preprocess_image(src.data_ptr(), h, w, int(image_size), out[k].data_ptr(), stream),
                       "pd_preprocess_image")
            bboxes.append(bbox)
            scales.append(scale)
    # assume all the images have the same shape for GGS   (:46-47)
    image_info = {"size": (min_hw, min_hw), "bboxes_xyxy": np.stack(bboxes), "resized_scales": np.stack(scales)}
    return out, image_info
