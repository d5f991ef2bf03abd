"""The keypoint bookkeeping of pose_diffusion/util/match_extraction.py.

`colmap_keypoint_to_pytorch3d` (:50-77) maps COLMAP keypoints of the original images into the cropped + resized frames
that the sampler sees and emits the (kp1, kp2, i12) arrays `pd_ggs_set_matches` takes; it is host numpy in the reference
and stays host numpy here (a few thousand points, once per sequence).  Match extraction itself (`extract_match`,
`run_hloc`: SuperPoint + SuperGlue through hloc / pycolmap) is outside this engine's scope."""
import numpy as np


def colmap_keypoint_to_pytorch3d(matches, keypoints, image_info):
    kp1, kp2, i12 = [], [], []
    bbox_xyxy, scale = image_info["bboxes_xyxy"], image_info["resized_scales"]
    keypoints = dict(keypoints)
    for idx in keypoints:
        cur = keypoints[idx] - 0.5                                            # COLMAP -> OpenCV pixel centres   (:56)
        cur = cur - [bbox_xyxy[idx - 1][0], bbox_xyxy[idx - 1][1]]            # into the crop; COLMAP ids start at 1   (:60)
        keypoints[idx] = cur * scale[idx - 1]                                 # into the resized frame   (:61)
    for (r_idx, q_idx), pair_match in matches.items():
        if pair_match is not None:
            kp1.append(keypoints[r_idx][pair_match[:, 0]])
            kp2.append(keypoints[q_idx][pair_match[:, 1]])
            i12.append(np.repeat(np.array([[r_idx - 1, q_idx - 1]]), len(pair_match), axis=0))
    if kp1:
        kp1, kp2, i12 = map(np.concatenate, (kp1, kp2, i12), (0, 0, 0))
    else:
        kp1 = kp2 = i12 = None
    return kp1, kp2, i12


def extract_match(image_paths=None, image_folder_path=None, image_info=None):
    raise ImportError("extract_match needs hloc + pycolmap (SuperPoint / SuperGlue), which this engine does not replace; "
                      "run the reference's util/match_extraction.py for the matches and pass them as matches_dict")
