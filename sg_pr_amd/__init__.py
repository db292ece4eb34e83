"""sg_pr_amd - MI355X-native semantic-graph pair scorer (drop-in for the SG_PR hot path)."""
__version__ = "0.1.0"
