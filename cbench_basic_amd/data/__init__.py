from .image_folder import ImageFolderDataset, RandomImageDataset, batched  # noqa: F401
