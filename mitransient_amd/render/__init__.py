from .transient_image_block import TransientImageBlock

__all__ = ["TransientImageBlock"]
