from .omni3d_evaluation import Omni3DEvaluator, inference_on_dataset, instances_to_coco_json  # noqa: F401
