"""Stand-in models for the object-reasoning loop tests: both read their answer back out of the crop (unmore_amd.synth.reasoning_scene
puts object-like fields INTO the image's channels), so the reference's `Object_Discovery` on the CPU (tests/golden/
make_golden_r6_discovery.py) and unmore_amd's on the GPU run the same "networks" on their own crops.  torch only; no parameters."""
import torch


class FieldsFromCrop(torch.nn.Module):
    """objectness-net stand-in: sdf_maps = channel 0 of the crop, center_fields = channels 1, 2"""
    def forward(self, images):
        return {"center_fields": images[:, 1:3].contiguous(), "sdf_maps": images[:, 0:1].contiguous()}

    def get_prediction(self, images):
        return self.forward(images)


class ObjectFraction(torch.nn.Module):
    """existence-classifier stand-in: the fraction of the crop that lies inside an object, [B, 1]"""
    def forward(self, images):
        return (images[:, 0] > 0).to(torch.float32).mean(dim=(1, 2)).unsqueeze(1)
