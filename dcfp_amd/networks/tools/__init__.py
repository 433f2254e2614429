"""Head building blocks of the segmentation networks (ASPP)."""
