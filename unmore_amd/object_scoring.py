"""Host-side mirror of the reference's `Object_Scoring` (object_scoring.py:43-272) -- the second caller of the ObjectnessNet hot path at
inference: for the boxes object discovery found in an image, predict both fields and the existence score on their 128 x 128 crops, turn
the fields into a mask per box, tighten the box to its mask, NMS, and score what is left.  What is NOT mirrored: the COCO dataset
object, the raw-annotation JSON, the pycocotools RLE string of a mask (`score_image` returns the masks themselves).

The reference builds, per proposal, two image-sized int64 canvases in a Python loop (Resize of the crop-sized mask into the box, paste),
stacks them ([N,H,W] twice, plus their union), takes every tight box through pycocotools on the host, and only then runs NMS.  Here one
launch reduces each proposal's pasted union mask to its tight box, its area and the two field maxima without writing anything
image-sized (csrc/reasoning.hip::mask_paste_stats_kernel); NMS runs on those; the masks are materialised for the survivors only."""

import numpy as np
import torch

from . import _lib as L
from . import reasoning
from .ops import _p, _stream


class Object_Scoring:
    def __init__(self, args, device, objectness_model=None, binary_classifier_model=None):
        """object_scoring.py:45-104: both networks restored from `args.objectness_resume` / `args.binary_classifier_resume` (:59-90)
        unless they are handed in; fp32, eval, frozen."""
        self.args, self.device = args, torch.device(device)
        if objectness_model is None:
            from .objectness_net import ObjectnessNet
            objectness_model = ObjectnessNet(device=self.device, image_size=args.image_size, backbone_type=args.backbone_type, args=args).to(self.device)
            objectness_model.load_state_dict(torch.load(args.objectness_resume, map_location=self.device)["model_state_dict"], strict=True)
            objectness_model = objectness_model.to(torch.float32)
        if binary_classifier_model is None:
            from .binary_classifier import Binary_Classifier
            binary_classifier_model = Binary_Classifier(device=self.device, image_size=args.image_size, args=args).to(self.device)
            binary_classifier_model.load_state_dict(torch.load(args.binary_classifier_resume, map_location=self.device)["model_state_dict"], strict=True)
            binary_classifier_model = binary_classifier_model.to(torch.float32)
        self.objectness_model, self.binary_classifier_model = objectness_model, binary_classifier_model
        for m in (objectness_model, binary_classifier_model):
            if isinstance(m, torch.nn.Module):
                m.eval()
                for p in m.parameters():
                    p.requires_grad = False

    def get_prediction_with_proposals(self, image, proposals):
        """object_scoring.py:112-157: {'pred_boundary_fields' [N,128,128], 'pred_center_fields' [N,2,128,128], 'pred_existence_scores' [N]}"""
        out = reasoning.get_prediction_with_proposals(self.objectness_model, self.binary_classifier_model, image, proposals)
        return {k: out[k] for k in ("pred_boundary_fields", "pred_center_fields", "pred_existence_scores")}

    def score_image(self, image, raw_proposals):
        """object_scoring.py:182-255 for one image.  image [3,H,W] f32 (moved to the GPU), raw_proposals: N boxes [x1,y1,x2,y2].
        Returns a dict of what the reference writes per surviving box, in NMS order: 'tight_bboxes' [K,4] f32 (x1,y1,x2,y2), 'masks'
        [K,H,W] u8 on the GPU, and numpy arrays 'score' (f64), 'existence_score', 'center_score', 'boundary_score' (f32), 'area_score' (f64)."""
        image = image.to(self.device, torch.float32)
        H, W = image.shape[-2], image.shape[-1]
        props = torch.as_tensor(np.asarray(raw_proposals, dtype=np.float64)).reshape(-1, 4)
        N = len(props)
        if N == 0:
            return None
        pred = self.get_prediction_with_proposals(image, props)
        sdf = pred["pred_boundary_fields"].contiguous().float()
        cen = pred["pred_center_fields"].contiguous().float()
        S = sdf.shape[-1]
        # the box a mask is pasted into: floor / ceil corners (:200-202), clipped like the slice `canvas[y1:y2, x1:x2]` clips them
        ib = torch.stack([torch.floor(props[:, 0]), torch.floor(props[:, 1]), torch.ceil(props[:, 2]), torch.ceil(props[:, 3])], 1).to(torch.int32)
        ib[:, 0].clamp_(0, W); ib[:, 2].clamp_(0, W); ib[:, 1].clamp_(0, H); ib[:, 3].clamp_(0, H)   # noqa: E702
        ib = ib.to(self.device).contiguous()
        stats = torch.empty((N, 5), dtype=torch.int32, device=self.device)
        maxima = torch.empty((N, 2), dtype=torch.float32, device=self.device)
        L.check(L.lib().umr_mask_paste_stats(_p(sdf), _p(cen), _p(ib), N, S, H, W, _p(stats), _p(maxima), _stream()), "umr_mask_paste_stats")
        tight = stats[:, :4].to(torch.float32)                                   # torch.FloatTensor(tight_bboxes), :235
        max_center, max_boundary = maxima[:, 0], maxima[:, 1]                    # :189-193
        keep = reasoning.nms(tight, max_boundary, iou_threshold=0.5)            # :238
        K = len(keep)
        masks = torch.empty((K, H, W), dtype=torch.uint8, device=self.device)
        L.check(L.lib().umr_mask_paste(_p(sdf), _p(cen), _p(ib), _p(keep.contiguous()), K, S, H, W, _p(masks), _stream()), "umr_mask_paste")
        area = stats[:, 4][keep].cpu().numpy().astype(np.int64)                  # final_binary_masks.sum(1).sum(1), :244-245
        existence = pred["pred_existence_scores"][keep].cpu().numpy()
        center = max_center[keep].cpu().numpy()
        boundary = max_boundary[keep].cpu().numpy()
        mask_scores = area / area.max()
        area_score = np.power(mask_scores, 0.25)
        score = existence * center * boundary * area_score                      # :255
        return {"tight_bboxes": tight[keep], "masks": masks, "score": score, "existence_score": existence, "center_score": center,
                "boundary_score": boundary, "area_score": area_score, "keep": keep}

    def annotations(self, image_id, scored):
        """the reference's per-box records (:257-267) without 'segmentation' (a pycocotools RLE string there; `scored['masks']` here)"""
        out = []
        if scored is None:
            return out
        for i, (x1, y1, x2, y2) in enumerate(scored["tight_bboxes"].cpu().numpy()):
            out.append({"image_id": image_id, "category_id": 1, "score": scored["score"][i], "bbox": [x1, y1, x2 - x1, y2 - y1],
                        "existence_score": scored["existence_score"][i], "center_score": scored["center_score"][i],
                        "boundary_score": scored["boundary_score"][i], "area_score": scored["area_score"][i]})
        return out

    def main_object_scoring(self, images, raw_annotations):
        """object_scoring.py:172-272 without the dataset object and the JSON files: `images` yields (image_id, image), `raw_annotations`
        maps str(image_id) -> boxes (the discovery results); returns the annotation list"""
        out = []
        for image_id, image in images:
            if str(image_id) not in raw_annotations:
                continue
            out.extend(self.annotations(image_id, self.score_image(image, raw_annotations[str(image_id)])))
        return out
