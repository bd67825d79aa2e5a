"""Import-path aliases: `from src.models import piv_liteflownet` / `from src.correlation import FunctionCorrelation`
keep working for code written against the reference's layout (put piv_liteflownet-pytorch_amd/ on sys.path)."""
