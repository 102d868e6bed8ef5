"""COCO-json side of the detection validation (plain host code fed by the device post-process).

  * invert_affine          -- DetectionHeader.invert_affine (head_detect/detection.py:217-229): boxes from network-input to source-image scale
  * detections_to_coco     -- the result records HydraTrainer.valid builds (train.py:335-364): x1,y1,x2,y2 -> x,y,w,h, category_id = class + 1
  * write_results          -- val_bbox_results.json as train.py:416-421 writes it (the file COCO().loadRes / COCOeval consume)
  * coco_ground_truth      -- the ground-truth dataset dict of head_detect/gen_val_json.py:4-117 from label records (the reference reads the
                              image size with cv2.imread; here the caller passes it, or PIL reads it)
pycocotools itself (COCOeval) is third party and absent from this image: the files written here are its inputs.
"""
from __future__ import annotations

import json
import os
from typing import Iterable, List, Optional, Sequence

import numpy as np

CATEGORIES = [("road", "roadtext"), ("person", "pedestrian"), ("road", "guidearrow"), ("traffic", "traffic"), ("obstacle", "obstacle"),
              ("vehicle", "vehicle_wheel"), ("road", "roadsign"), ("vehicle", "vehicle"), ("traffic", "vehicle_light")]


def invert_affine(metas, preds: List[dict]) -> List[dict]:
    """metas: a float scale, or per image (new_w, new_h, old_w, old_h, padding_w, padding_h); preds: the per-image dicts of
    detectheader.decode (rois [K,4] x1,y1,x2,y2 in network-input pixels) -- rescaled IN PLACE like the reference"""
    for i in range(len(preds)):
        if len(preds[i]["rois"]) == 0:
            continue
        if isinstance(metas, float):
            preds[i]["rois"][:, [0, 2]] = preds[i]["rois"][:, [0, 2]] / metas
            preds[i]["rois"][:, [1, 3]] = preds[i]["rois"][:, [1, 3]] / metas
        else:
            new_w, new_h, old_w, old_h, _, _ = metas[i]
            preds[i]["rois"][:, [0, 2]] = preds[i]["rois"][:, [0, 2]] / (new_w / old_w)
            preds[i]["rois"][:, [1, 3]] = preds[i]["rois"][:, [1, 3]] / (new_h / old_h)
    return preds


def detections_to_coco(preds: Sequence[dict], first_image_id: int) -> List[dict]:
    """train.py:335-364: one record per kept box; image ids count from `first_image_id` (= iter_idx * batch_size_valid + 1)"""
    out = []
    for k, pr in enumerate(preds):
        rois = np.asarray(pr["rois"], dtype=np.float32)
        if rois.ndim != 2 or rois.shape[0] == 0:
            continue
        rois = rois.copy()
        rois[:, 2] -= rois[:, 0]
        rois[:, 3] -= rois[:, 1]
        for r in range(rois.shape[0]):
            out.append({"image_id": first_image_id + k, "category_id": int(pr["class_ids"][r]) + 1, "score": float(pr["scores"][r]),
                        "bbox": rois[r, :].tolist()})
    return out


def write_results(records: List[dict], eval_dir: str, name: str = "val_bbox_results.json") -> Optional[str]:
    """train.py:412-421: nothing is written when the model produced no detection"""
    if not records:
        return None
    os.makedirs(eval_dir, exist_ok=True)
    path = os.path.join(eval_dir, name)
    if os.path.exists(path):
        os.remove(path)
    with open(path, "w") as f:
        json.dump(records, f, indent=4)
    return path


def _image_size(path):
    from PIL import Image
    with Image.open(path) as im:
        w, h = im.size
    return h, w


def coco_ground_truth(images: Iterable[dict]) -> dict:
    """gen_coco_label's dataset dict (gen_val_json.py:4-117).  images: dicts {file_name, annos: rows "x1,y1,x2,y2,category" (str or
    sequence), [height, width]}; images without annotations are skipped and do not consume an id, exactly like the reference's loop."""
    ds = {"info": {"description": "This is stable 1.0 version of the 2014 MS COCO dataset.", "url": "http://mscoco.org", "version": "1.0", "year": 2021,
                   "contributor": "Group", "date_created": "2021-09-01 11:35:00.000000"},
          "images": [], "annotations": [],
          "categories": [{"supercategory:": sc, "id": i + 1, "name": nm} for i, (sc, nm) in enumerate(CATEGORIES)]}
    cnt = annoid = 0
    for rec in images:
        annos = rec["annos"]
        if len(annos) == 0:
            continue
        cnt += 1
        if "height" in rec and "width" in rec:
            height, width = rec["height"], rec["width"]
        else:
            height, width = _image_size(rec["file_name"])
        ds["images"].append({"license": 5, "file_name": rec["file_name"], "coco_url": "local", "height": height, "width": width,
                             "date_captured": "2018_08_29 10:10:10", "flickr_url": "local", "id": cnt})
        for a in annos:
            parts = a.strip("\n").split(",") if isinstance(a, str) else list(a)
            x1, y1, x2, y2 = (float(v) for v in parts[:4])
            category = int(parts[4])
            wid, hei = max(0, int(x2 - x1)), max(0, int(y2 - y1))
            annoid += 1
            ds["annotations"].append({"segmentation": [], "iscrowd": 0, "area": wid * hei, "image_id": cnt, "bbox": [x1, y1, wid, hei],
                                      "category_id": category, "id": annoid})
    return ds


def gen_coco_label(root_dir: str, list_name: str = "valid.txt") -> str:
    """gen_val_json.py:29-117 on a dataset tree (<root>/list/<list_name> of image paths, labels in labels_object/*.txt): writes
    <root>/eval_detect/gt_bbox_results.json once and returns its path"""
    target = os.path.join(root_dir, "eval_detect")
    os.makedirs(target, exist_ok=True)
    json_name = os.path.join(target, "gt_bbox_results.json")
    if os.path.exists(json_name):
        return json_name
    recs = []
    for line in open(os.path.join(root_dir, "list", list_name)).readlines():
        img = line.strip("\n")
        with open(img.replace("images", "labels_object").replace(".jpg", ".txt")) as f:
            recs.append({"file_name": img, "annos": f.readlines()})
    with open(json_name, "w") as f:
        json.dump(coco_ground_truth(recs), f)
    return json_name
