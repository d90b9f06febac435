from .aspect_ratio_bucket import AspectRatioBucketConfig, AspectRatioBucketManager, generate_buckets
from .bucket import Bucket, BucketDataset
from .latent_cache import CachedLatentDatasetConfig, LatentCacheWriter
from .synthetic import SyntheticLatentBucket, SyntheticLatentDatasetConfig
from .util import DatasetConfig

__all__ = ["AspectRatioBucketConfig", "AspectRatioBucketManager", "generate_buckets", "Bucket", "BucketDataset",
           "SyntheticLatentBucket", "SyntheticLatentDatasetConfig", "DatasetConfig", "CachedLatentDatasetConfig", "LatentCacheWriter"]
