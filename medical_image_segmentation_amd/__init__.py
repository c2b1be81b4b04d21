"""MI355X-native U-Net training hot path behind the reference's own call signatures.

    from medical_image_segmentation_amd.model.unet import UNet
    from medical_image_segmentation_amd.model.a_unet import A_UNet
    from medical_image_segmentation_amd.utils.loss_fn import diceloss
    from medical_image_segmentation_amd.utils.metrics import Metrics
    from medical_image_segmentation_amd.utils.train_and_eval import train_one_epoch, evaluate
    from medical_image_segmentation_amd.utils.slide_predict import SlidingWindowPredictor

mirror `model/unet.py`, `model/a_unet.py`, `utils/loss_fn.py`, `utils/metrics.py`, `utils/train_and_eval.py` and
`utils/slide_predict.py` of Runner-xc/Medical_image_segmentation.  All arithmetic runs in
hand-written gfx950 kernels loaded from libmisamd.so (see include/misamd.h).
"""
__version__ = "0.1.0"
