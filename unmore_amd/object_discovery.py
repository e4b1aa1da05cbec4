"""Host-side mirror of the reference's `Object_Discovery` (object_reasoning.py:43-665) -- the caller of the ObjectnessNet hot path at
inference: proposal grid -> existence checking -> centre reasoning (split the boxes that hold more than one centre) -> boundary
reasoning (up to `n_round` rounds of "crop, predict the boundary-distance map, move the box edges") -> NMS.  Same class name, method
names, argument meaning and return dictionaries; what is NOT mirrored: the COCO dataset object, the results folder and the JSON dump
(`main_object_discovery` takes the images and returns the dictionary the reference would have written).

Design (MI355X-first, not a transliteration): the reference walks every box in Python (crop, Resize, list append, per-box tensors) and
indexes with boolean masks after every step, each a host synchronisation.  Here a round is a handful of launches over ALL boxes -- one
crop+resize kernel per batch (csrc/reasoning.hip), the net, one boundary-delta kernel, masked element-wise box arithmetic on the
device -- with three host synchronisations per round and image (the survivors, the distinct crops, the crops not seen before) where the
reference has several per BOX.  Things the reference's loop does not have:
  * the boundary rounds read `sdf_maps` only, so the centre head -- most of a 128x128 crop's forward -- is not evaluated
    (`ObjectnessNet.get_prediction(heads=("sdf_maps",))`);
  * a box that a round labels "good" (1) and leaves exactly where it was is a fixed point of the round (same crop, same map, same
    label, zero delta), so it is carried through the remaining rounds without being evaluated again; when every box is such a fixed
    point the remaining rounds are skipped.  Results are what the reference's fifty full rounds produce.
  * what the net sees of a box is its crop, cut at floor / ceil of the corners: boxes that share those four integers share the crop and
    everything computed from it, so each DISTINCT crop is evaluated once per image (late rounds hold hundreds of boxes clustered on
    a few objects, and the boxes that keep the loop running to its fiftieth round oscillate between a few positions: after their
    second visit they are looked up, not evaluated).
(And a batch is 200 crops in the boundary rounds, not 50: `boundary_batch`.)
Arithmetic that decides something (thresholds, label rules, the order of operations in the box updates, dtypes: float64 proposals on
the first round, float32 afterwards) follows the reference line by line; each method cites its lines."""

import numpy as np
import torch

from . import reasoning

_KEY = 1 << 15            # integer corners < 32768 pack into one int64 key
BOUNDARY_BATCH = 200      # crops per net call in the boundary rounds (the reference: 50, object_reasoning.py:397): dpt_large's sdf-only forward
                          # does 2 813 crops/s at 50 and 3 317 at 200 in fp32, 7 966 and 13 388 in bf16 (tools/probe/sdf_only_batch.py)
_DEFAULTS = dict(class_score_thres=0.1, center_score_max_thres=0.009, analyze_cc=False, max_sdf_thres=0.5, max_shrink_threshold=16,
                 delta_ratio=0.5, n_round=50, proposal_area_thres=50, image_size=128)       # object_reasoning.py:701-710,681


class Object_Discovery:
    def __init__(self, args, device, objectness_model=None, binary_classifier_model=None):
        """object_reasoning.py:44-107.  args: the reference's namespace (missing reasoning hyper-parameters take the reference's
        defaults).  The two models are built and restored from `args.objectness_resume` / `args.binary_classifier_resume` as the
        reference does (:58-88) unless they are handed in."""
        self.args = args
        self.device = torch.device(device)
        for k, v in _DEFAULTS.items():
            if not hasattr(args, k):
                setattr(args, k, v)
        if objectness_model is None:
            from .objectness_net import ObjectnessNet
            objectness_model = ObjectnessNet(device=self.device, image_size=args.image_size, backbone_type=args.backbone_type, args=args)
            objectness_model = objectness_model.to(self.device)
            ckpt = torch.load(args.objectness_resume, map_location=self.device)
            objectness_model.load_state_dict(ckpt["model_state_dict"], strict=True)
            objectness_model = objectness_model.to(torch.float32)
        if binary_classifier_model is None:
            from .binary_classifier import Binary_Classifier
            binary_classifier_model = Binary_Classifier(device=self.device, image_size=args.image_size, args=args).to(self.device)
            ckpt = torch.load(args.binary_classifier_resume, map_location=self.device)
            binary_classifier_model.load_state_dict(ckpt["model_state_dict"], strict=True)
            binary_classifier_model = binary_classifier_model.to(torch.float32)
        self.objectness_model = objectness_model
        self.binary_classifier_model = binary_classifier_model
        for m in (self.objectness_model, self.binary_classifier_model):
            if isinstance(m, torch.nn.Module):
                m.eval()
                for p in m.parameters():
                    p.requires_grad = False
        self.stats = {}       # per image: rounds run, crops evaluated (tools / tests read it; the reference prints counts instead)
        self.boundary_batch = int(getattr(args, "boundary_batch", BOUNDARY_BATCH))
        self.pipelined_center_sweep = True   # centre reasoning through reasoning.sweep_proposals when the net is unmore_amd's own
        self.share_equal_crops = True      # boundary rounds: boxes with the same integer corners are cropped and evaluated once ...
        self.remember_crops = True         # ... per IMAGE, not per round: a crop evaluated in an earlier round is looked up
        self.carry_fixed_points = True     # False: every surviving box is evaluated in every round, as the reference does (tests compare)

    # ------------------------------------------------------------------ static helpers
    @staticmethod
    def generate_random_proposal(height, width):
        """object_reasoning.py:109-137: for grid sizes 32 ... 512 a square (2g x 2g), a tall (g x 2g) and a wide (2g x g) anchor around
        every grid point (x fastest, the three shapes innermost), clipped to the image, plus the whole image.  float64 [N, 4]."""
        out = []
        for g in (32, 64, 128, 256, 512):
            ys = np.arange(0, height, g, dtype=int)
            xs = np.arange(0, width, g, dtype=int)
            cx, cy = np.meshgrid(xs, ys)
            c = np.stack([cx.ravel(), cy.ravel(), cx.ravel(), cy.ravel()], axis=1).astype(np.float64)          # [P, 4]
            base = np.array([[-g, -g, g, g], [-g / 2, -g, g / 2, g], [-g, -g / 2, g, g / 2]], dtype=np.float64)
            out.append((c[:, None, :] + base[None, :, :]).reshape(-1, 4))
        out = np.concatenate(out, axis=0)
        out[:, 0] = np.where(out[:, 0] < 0, 0, out[:, 0])
        out[:, 1] = np.where(out[:, 1] < 0, 0, out[:, 1])
        out[:, 2] = np.where(out[:, 2] >= width, width, out[:, 2])
        out[:, 3] = np.where(out[:, 3] >= height, height, out[:, 3])
        return np.concatenate((out, np.array([[0, 0, width, height]], dtype=np.float64)), axis=0)

    @staticmethod
    def update_bbox_with_boundary_fields(sdf_maps):
        """object_reasoning.py:139-174 on the device (one launch): (delta_x1, delta_y1, delta_x2, delta_y2), each [B]"""
        return reasoning.update_bbox_with_boundary_fields(sdf_maps)

    @staticmethod
    def post_process_bbox_update(original_bboxes, delta_bboxes, delta_scale_x=128, delta_scale_y=128):
        """object_reasoning.py:176-197: deltas measured on the 128 x 128 map, scaled to the box's own extent.  The result has the
        boxes' dtype (the column assignments of the reference cast back to it)."""
        x_ratio = (original_bboxes[:, 2] - original_bboxes[:, 0]) / delta_scale_x
        y_ratio = (original_bboxes[:, 3] - original_bboxes[:, 1]) / delta_scale_y
        ratio = torch.stack([x_ratio, y_ratio, x_ratio, y_ratio], dim=1)
        return (original_bboxes + delta_bboxes * ratio).to(original_bboxes.dtype)

    @staticmethod
    def unravel_index(index, shape):
        """object_reasoning.py:198-204"""
        coords = [None] * len(shape)
        for axis in range(len(shape) - 1, -1, -1):      # last axis varies fastest
            coords[axis] = index % shape[axis]
            index = index // shape[axis]
        return tuple(coords)

    @staticmethod
    def separate_connected_components(binary_masks):
        """object_reasoning.py:206-257 (--analyze_cc, README.md:176): 8-connected components of every mask (scipy.ndimage.label with a
        3 x 3 structure there; csrc/reasoning.hip::mask_components_kernel here, same numbering); boxes [x1, y1, x2, y2) of the components
        of the masks that have several, and per mask whether it has exactly one.  binary_masks [B,S,S] on the GPU."""
        m = binary_masks.to(torch.float32)
        counts, boxes = reasoning.mask_components(torch.where(m > 0, 1.0, -1.0), torch.zeros((m.shape[0], 2) + tuple(m.shape[1:]), device=m.device))
        return Object_Discovery._components_to_lists(counts, boxes)

    @staticmethod
    def _components_to_lists(counts, boxes):
        counts_h = counts.cpu()
        if int(counts_h.max()) > boxes.shape[1]:
            raise RuntimeError(f"separate_connected_components: a mask has {int(counts_h.max())} components, more than the {boxes.shape[1]} the kernel records")
        boxes_h = boxes.cpu()
        combined = {"single": [boxes_h[b, 0].tolist() for b in torch.nonzero(counts_h == 1).flatten().tolist()], "multi": []}
        for b in torch.nonzero(counts_h > 1).flatten().tolist():
            combined["multi"].extend(boxes_h[b, :int(counts_h[b])].tolist())
        return combined, (counts_h == 1).to(torch.int64).tolist()

    @staticmethod
    def enlarge_proposals(proposals, image_shape, ratio):
        """object_reasoning.py:259-291: scale every box about its centre, clip to the image, truncate to int"""
        height, width = image_shape
        out = []
        for x1, y1, x2, y2 in proposals:
            cx, cy = (x1 + x2) / 2, (y1 + y2) / 2
            nw, nh = (x2 - x1) * ratio, (y2 - y1) * ratio
            out.append([int(max(cx - nw / 2, 0)), int(max(cy - nh / 2, 0)), int(min(cx + nw / 2, width)), int(min(cy + nh / 2, height))])
        return out

    def filter_small_proposal(self, proposals, labels):
        """object_reasoning.py:293-299"""
        keep = (proposals[:, 2] - proposals[:, 0]) * (proposals[:, 3] - proposals[:, 1]) > self.args.proposal_area_thres
        return proposals[keep], labels[keep]

    # ------------------------------------------------------------------ the net on proposal crops
    def _predict(self, crops, heads=None):
        m = self.objectness_model
        with torch.no_grad():
            if heads is not None and hasattr(m, "_HEAD_OF"):          # unmore_amd.ObjectnessNet: evaluate the requested heads only
                return m.get_prediction(crops, heads=heads)
            return m(crops)

    def get_prediction_with_proposals(self, proposals, image):
        """object_reasoning.py:301-337: (sdf_maps [N,128,128], center_fields [N,2,128,128]) of the proposals' 128 x 128 crops, 50 per batch"""
        sdf, cen = [], []
        for i in range(0, len(proposals), 50):
            crops, _ = reasoning.crop_resize(image, proposals[i:i + 50], 128)
            pred = self._predict(crops.to(torch.float32))
            sdf.append(pred["sdf_maps"].squeeze(1))
            cen.append(pred["center_fields"])
        return torch.cat(sdf, dim=0), torch.cat(cen, dim=0)

    def center_field_to_anti_center_map(self, vote_maps, kernel_size=5):
        """object_reasoning.py:360-377 (float64 5 x 5 correlation / 24); evaluated on the device inside `center_reasoning`'s kernel --
        this method exists for callers that want the map itself"""
        assert kernel_size == 5
        B, _, H, W = vote_maps.shape
        ones = torch.ones((B, H, W), dtype=torch.float32, device=vote_maps.device)
        _, _, sc = reasoning.center_peaks(ones, vote_maps, border=0, erode_rounds=0, return_scores=True)
        return sc

    def existence_checking(self, image, proposals):
        """object_reasoning.py:491-523"""
        return reasoning.existence_checking(self.binary_classifier_model, image, proposals, num_img_per_batch=128)

    # ------------------------------------------------------------------ centre reasoning
    def center_reasoning(self, image, proposals):
        """object_reasoning.py:525-580.  Union mask, three 9 x 9 erosions, anti-centre score, 10-pixel border, per-map maximum and first
        flat argmax are ONE launch (reasoning.center_peaks); a box whose maximum exceeds `center_score_max_thres` holds more than one
        centre and is replaced by its left / right / top / bottom parts at the peak (in that order, box after box)."""
        a = self.args
        proposals = torch.as_tensor(proposals).to(self.device)
        m = self.objectness_model
        if self.pipelined_center_sweep and hasattr(m, "_HEAD_OF") and hasattr(m, "compute_dtype") and not a.analyze_cc:
            # unmore_amd's own net: the batches of 50 go round three HIP streams, and in fp32 the sweep is certificate-driven -- three-term
            # products first, six-term only for the proposals whose peak index / side of the threshold is not PROVABLY the six-term one
            # (reasoning.sweep_proposals: 1.7x on a 1 225-proposal image, same indices and decisions; bench.py --workload cfg5)
            mx, am, _ = reasoning.sweep_proposals(self.objectness_model, image, proposals, precision="certified")
            H = W = 128
            sdf_maps = center_fields = None
        else:
            sdf_maps, center_fields = self.get_prediction_with_proposals(proposals, image)
            mx, am = reasoning.center_peaks(sdf_maps, center_fields)
            H, W = sdf_maps.shape[-2], sdf_maps.shape[-1]
        fail = mx > a.center_score_max_thres
        passed = proposals[~fail]
        split = []
        if bool(fail.any()):
            pf, idx = proposals[fail], am[fail]
            # (y, x) of the peak as fractions of the map: an int64 tensor divided by a Python int is float32 in torch (:553-554), and
            # the box arithmetic promotes to the proposals' dtype
            y_ratio = ((idx // W) / H).to(torch.float32)
            x_ratio = ((idx % W) / W).to(torch.float32)
            x1, y1, x2, y2 = pf[:, 0], pf[:, 1], pf[:, 2], pf[:, 3]
            xs = x1 + (x2 - x1) * x_ratio
            ys = y1 + (y2 - y1) * y_ratio
            left = torch.stack([x1, y1, xs, y2], dim=1)
            right = torch.stack([xs, y1, x2, y2], dim=1)
            top = torch.stack([x1, y1, x2, ys], dim=1)
            bottom = torch.stack([x1, ys, x2, y2], dim=1)
            # (torch.tensor([...]) of 0-dim tensors keeps their dtype, :555-558: float64 rows for the float64 proposal grid)
            split = torch.stack([left, right, top, bottom], dim=1).reshape(-1, 4)
        if a.analyze_cc:
            counts, cboxes = reasoning.mask_components(sdf_maps[~fail], center_fields[~fail])      # the union masks' components, on the device
            cc, _single = self._components_to_lists(counts, cboxes)
            multi = self.enlarge_proposals(cc["multi"], (self.height, self.width), ratio=1.5)
            extra = torch.tensor(multi, dtype=torch.float32, device=self.device).reshape(-1, 4)
            # (the reference concatenates onto its split list and fails when that list is empty, :571; here the extra boxes stand alone then)
            split = torch.cat((split, extra), dim=0) if len(split) else extra
        return {"proposals_pass_singularity": passed, "splited_new_proposals": split}

    # ------------------------------------------------------------------ boundary reasoning
    def _round_plan(self, image, proposals, memo):
        """first third of a round: which crops of this image have to go through the net (one host synchronisation: torch.unique)"""
        H, W = image.shape[-2], image.shape[-1]
        # what the net sees of a box is its crop, and the crop is cut at floor / ceil of the corners (:404): boxes that share those
        # four integers share crop, map, maximum and deltas.  Late rounds hold hundreds of boxes clustered on a few objects -- the net
        # runs once per DISTINCT crop and the per-box arithmetic of _round_finish picks its crop's results up
        plan = dict(image=image, proposals=proposals, H=H, W=W, inv=None, hit=None, pos=None, ukey=None, new_key=None, eval_boxes=proposals)
        if self.share_equal_crops and len(proposals) > 1:
            b64 = proposals.detach().to(torch.float64)
            corners = torch.stack([torch.floor(b64[:, 0]), torch.floor(b64[:, 1]), torch.ceil(b64[:, 2]), torch.ceil(b64[:, 3])], 1).to(torch.int64)
            corners = torch.minimum(corners.clamp_(min=0), torch.tensor([W, H, W, H], device=corners.device))      # as the crop clips them
            key = ((corners[:, 0] * _KEY + corners[:, 1]) * _KEY + corners[:, 2]) * _KEY + corners[:, 3]           # one int64 per crop
            ukey, inv = torch.unique(key, return_inverse=True)
            new_key = ukey
            if memo is not None and memo.get("keys") is not None and len(memo["keys"]):
                # ... and a crop evaluated in an EARLIER round of this image is not evaluated again either: boxes that oscillate between
                # a few positions -- the ones that keep the loop running to its fiftieth round -- cost nothing after their second visit
                pos = torch.searchsorted(memo["keys"], ukey).clamp_(max=len(memo["keys"]) - 1)
                hit = memo["keys"][pos] == ukey
                new_key = ukey[~hit]
                plan.update(pos=pos, hit=hit)
            eval_boxes = torch.stack([new_key // (_KEY ** 3), (new_key // (_KEY ** 2)) % _KEY, (new_key // _KEY) % _KEY, new_key % _KEY], 1).to(torch.float64)
            plan.update(inv=inv, ukey=ukey, new_key=new_key, eval_boxes=eval_boxes)
        self.stats["boundary_distinct_crops"] = self.stats.get("boundary_distinct_crops", 0) + len(plan["eval_boxes"])
        return plan

    def _round_eval(self, plans):
        """second third: the crops of ALL the plans (one image each) through the net together, `boundary_batch` crops per call -- several
        images in lock-step fill the batches that one image's late rounds leave nearly empty.  -> per plan [n, 9]: max sdf | four deltas |
        four edge flags"""
        crops, edge = [], []
        for p in plans:
            if len(p["eval_boxes"]):
                c, e = reasoning.crop_resize(p["image"], p["eval_boxes"], 128)
                crops.append(c)
                edge.append(e.to(self.device))
        if not crops:
            return [torch.zeros((0, 9), dtype=torch.float32, device=self.device) for _ in plans]
        crops = torch.cat(crops, dim=0) if len(crops) > 1 else crops[0]
        nb = self.boundary_batch
        sdf = [self._predict(crops[i:i + nb].to(torch.float32), heads=("sdf_maps",))["sdf_maps"].squeeze(1) for i in range(0, len(crops), nb)]
        sdf = torch.cat(sdf, dim=0) if len(sdf) > 1 else sdf[0]
        vals = torch.cat([torch.amax(sdf, dim=(1, 2)).to(torch.float32)[:, None],
                          torch.stack(reasoning.update_bbox_with_boundary_fields(sdf), dim=1),                           # :441
                          torch.cat(edge, dim=0).to(torch.float32)], dim=1)
        return list(torch.split(vals, [len(p["eval_boxes"]) for p in plans], dim=0))

    def _round_finish(self, plan, vals, memo):
        """last third: every box picks up its crop's results and does its own arithmetic -> (updated boxes f32 [N,4] (zeros where filtered
        out), labels f32 [N]: -1 filtered out / 0 keep updating / 1 good)"""
        a = self.args
        proposals, H, W, inv, hit = plan["proposals"], plan["H"], plan["W"], plan["inv"], plan["hit"]
        if inv is not None:
            if hit is not None:
                allv = torch.empty((len(plan["ukey"]), 9), dtype=torch.float32, device=self.device)
                allv[hit], allv[~hit] = memo["vals"][plan["pos"][hit]], vals
            else:
                allv = vals
            new_key = plan["new_key"]
            if memo is not None and len(new_key):
                mk = new_key if memo.get("keys") is None else torch.cat([memo["keys"], new_key])
                mv = vals if memo.get("vals") is None else torch.cat([memo["vals"], vals])
                order = torch.argsort(mk)
                memo["keys"], memo["vals"] = mk[order], mv[order]
            vals = allv[inv]
        max_sdf, dx1, dy1, dx2, dy2, on_edge = vals[:, 0], vals[:, 1], vals[:, 2], vals[:, 3], vals[:, 4], vals[:, 5:9]
        keep = max_sdf > a.max_sdf_thres                                                                                # :421-427
        signed = torch.stack([-dx1, -dy1, dx2, dy2], dim=1)                      # > 0 expands, < 0 shrinks          # :444-445
        signed = torch.where((signed > 0) & (on_edge == 1), 0, 1).to(torch.float32) * signed
        max_expansion, max_shrink = torch.amax(signed, dim=1), torch.amin(signed, dim=1)                                # :446-447
        good = (max_expansion <= 0) & (max_shrink >= -a.max_shrink_threshold)                                           # :450-452
        dx1 = dx1 - torch.abs(dx1) * a.delta_ratio                                                                       # :457-460
        dy1 = dy1 - torch.abs(dy1) * a.delta_ratio
        dx2 = dx2 + torch.abs(dx2) * a.delta_ratio
        dy2 = dy2 + torch.abs(dy2) * a.delta_ratio
        delta = torch.stack([dx1, dy1, dx2, dy2], dim=1)
        delta = torch.where(good[:, None], torch.zeros_like(delta), delta)                                              # :463
        upd = self.post_process_bbox_update(proposals, delta, delta_scale_x=128, delta_scale_y=128)                      # :466
        lo = torch.zeros((), dtype=upd.dtype, device=upd.device)
        upd = torch.stack([torch.maximum(upd[:, 0], lo), torch.maximum(upd[:, 1], lo),                                  # :468-471
                           torch.minimum(upd[:, 2], lo + W), torch.minimum(upd[:, 3], lo + H)], dim=1)
        out = torch.where(keep[:, None], upd.to(torch.float32), torch.zeros((), dtype=torch.float32, device=upd.device))   # :479
        labels = torch.where(keep, good.to(torch.float32), torch.full((), -1.0, device=upd.device))
        return out, labels

    def _round(self, image, proposals, memo=None):
        """one round for `proposals` [N,4] of one image.  memo: the dict in which boundary_reasoning keeps the results of the crops of ITS
        image (keys: sorted int64 corner codes, vals: [n, 9]); None = nothing is remembered across calls"""
        plan = self._round_plan(image, proposals, memo)
        return self._round_finish(plan, self._round_eval([plan])[0], memo)

    def optimize_one_image_single_round(self, image, proposals, labels):
        """object_reasoning.py:379-487 (the incoming `labels` are overwritten there too, :392)"""
        proposals = torch.as_tensor(proposals).to(self.device)
        out, lab = self._round(image, proposals)
        return {"updated_bboxes": out, "labels": lab}

    def boundary_reasoning(self, image, proposals, n_round=50):
        """object_reasoning.py:582-612 (like the reference, the loop runs `args.n_round` rounds).  Fixed points are carried, crops shared and
        remembered (module docstring)."""
        return self.boundary_reasoning_many([image], [proposals])[0]

    def boundary_reasoning_many(self, images, proposals_list):
        """boundary_reasoning for several images in lock-step: round r of every image that is still running is planned, then the crops
        of all of them go through the net TOGETHER, then every image finishes its round -- same results per image as one at a time (a
        crop's result does not depend on its batch), fuller batches where single images have a few dozen crops left"""
        a = self.args
        st = []
        for image, proposals in zip(images, proposals_list):
            cur = torch.as_tensor(proposals).to(self.device)
            st.append(dict(image=image, cur=cur, labels=torch.zeros(len(cur), device=self.device),
                           frozen=torch.zeros(len(cur), dtype=torch.bool, device=self.device), done=False, result=None,
                           memo={} if (self.remember_crops and self.share_equal_crops) else None))
        rounds = crops = 0
        self.stats["boundary_distinct_crops"] = 0
        for _ in range(a.n_round):
            live = []
            for s_ in st:
                if s_["done"]:
                    continue
                cur, labels, frozen = s_["cur"], s_["labels"], s_["frozen"]
                keep = (cur[:, 2] - cur[:, 0]) * (cur[:, 3] - cur[:, 1]) > a.proposal_area_thres                        # :598 / :293-299
                cur, labels, frozen = cur[keep], labels[keep], frozen[keep]
                s_.update(cur=cur, labels=labels, frozen=frozen)
                if len(cur) == 0:                                           # (the boolean index above is a host sync)
                    s_.update(done=True, result={"proposals": [], "labels": []})
                    continue
                if self.carry_fixed_points and bool(frozen.all()):
                    s_["done"] = True                                       # every box is a fixed point: the remaining rounds change nothing
                    continue
                s_["active"] = ~frozen
                s_["plan"] = self._round_plan(s_["image"], cur[s_["active"]], s_["memo"])
                live.append(s_)
            if not live:
                break
            rounds += 1
            for s_, vals in zip(live, self._round_eval([s_["plan"] for s_ in live])):
                cur, labels, frozen, active = s_["cur"], s_["labels"], s_["frozen"], s_["active"]
                out, lab = self._round_finish(s_["plan"], vals, s_["memo"])
                crops += int(active.sum())
                new = torch.zeros((len(cur), 4), dtype=torch.float32, device=self.device)
                new_lab = torch.empty(len(cur), dtype=torch.float32, device=self.device)
                new[active], new_lab[active] = out, lab
                new[frozen], new_lab[frozen] = cur[frozen].to(torch.float32), labels[frozen]
                # a fixed point of the round: labelled good and returned bit for bit where it came from (from the second round on the
                # boxes are float32 on both sides; a float64 box of the first round is frozen only if the cast did not move it)
                now_fixed = torch.zeros(len(cur), dtype=torch.bool, device=self.device)
                now_fixed[active] = (lab == 1) & (out.to(cur.dtype) == cur[active]).all(dim=1)
                if self.carry_fixed_points:
                    frozen = frozen | now_fixed
                s_.update(cur=new, labels=new_lab, frozen=frozen, plan=None)
        self.stats.update(boundary_rounds=rounds, boundary_crops=crops)
        return [s_["result"] if s_["result"] is not None else {"proposals": s_["cur"], "labels": s_["labels"]} for s_ in st]

    # ------------------------------------------------------------------ one image, all images
    def _before_boundary(self, image):
        """Steps 0-2 of main_object_discovery (:619-645) for one image: the proposals that go into boundary reasoning, or None"""
        a = self.args
        self.height, self.width = image.shape[-2], image.shape[-1]
        proposals = torch.tensor(self.generate_random_proposal(height=self.height, width=self.width)).to(self.device)   # Step 0
        scores = self.existence_checking(image, proposals)["existence_scores"]                                        # Step 1
        proposals = proposals[(scores >= a.class_score_thres).to(self.device)]
        if len(proposals) == 0:
            return None
        res = self.center_reasoning(image, proposals)                                                                   # Step 2
        passed, split = res["proposals_pass_singularity"], res["splited_new_proposals"]
        if len(split) > 0:      # (the reference stacks an empty list and raises when no box failed the singularity check, :639 -> :513)
            scores = self.existence_checking(image, split)["existence_scores"]
            split = split[(scores >= a.class_score_thres).to(self.device)]
        if len(split) > 0:
            res2 = self.center_reasoning(image, split)
            proposals = torch.cat((passed, res2["proposals_pass_singularity"]), dim=0)
        else:
            proposals = passed
        return proposals if len(proposals) else None

    @staticmethod
    def _after_boundary(res):
        """:651-662: keep the boxes labelled good, NMS"""
        proposals, labels = res["proposals"], res["labels"]
        if len(proposals) == 0:
            return None
        good = labels == 1
        proposals = proposals[good]
        if len(proposals) == 0:
            return None
        keep = reasoning.nms(proposals.to(torch.float32), labels[good], iou_threshold=0.5)                             # :661
        return proposals[keep]

    def discover_image(self, image):
        """the body of main_object_discovery's loop (object_reasoning.py:619-662) for one [3,H,W] image on the GPU: the discovered boxes
        [K,4] (float32, on the GPU), or None where the reference `continue`s"""
        return self.discover_images([image])[0]

    def discover_images(self, images):
        """discover_image for several images whose boundary rounds run in lock-step (boundary_reasoning_many): one result per image,
        each what discover_image returns for it"""
        self.stats = {}
        images = [im.to(self.device, torch.float32) for im in images]
        starts = [self._before_boundary(im) for im in images]
        idx = [i for i, p in enumerate(starts) if p is not None]
        out = [None] * len(images)
        if idx:
            for i, res in zip(idx, self.boundary_reasoning_many([images[i] for i in idx], [starts[i] for i in idx])):
                out[i] = self._after_boundary(res)
        return out

    def main_object_discovery(self, images, images_in_lock_step=1):
        """object_reasoning.py:615-665 without the dataset object and the JSON file: `images` yields (image_id, image [3,H,W]); returns
        {image_id: boxes as a numpy array} -- the dictionary the reference dumps to discovery_results.json.  images_in_lock_step > 1:
        that many consecutive images share their boundary rounds' net calls (discover_images)."""
        results, group = {}, []

        def flush():
            for (image_id, _), boxes in zip(group, self.discover_images([im for _, im in group])):
                if boxes is not None:
                    results[image_id] = boxes.cpu().numpy()
            group.clear()
        for item in images:
            group.append(item)
            if len(group) >= max(1, int(images_in_lock_step)):
                flush()
        if group:
            flush()
        return results
