"""k-NN graph construction: the oracle on a hand-checkable case (CPU), the device kernel bit for bit (GPU)."""
import pytest
import torch

from oracle import reference_oracle as ro


def test_oracle_knn_known_answer():
    # two clouds of 4 collinear points; nearest neighbours are obvious, ties go to the lower index
    pos = torch.tensor([[0., 0, 0], [1, 0, 0], [3, 0, 0], [7, 0, 0],
                        [0, 0, 0], [0, 2, 0], [0, 4, 0], [0, 5, 0]])
    ei, ea = ro.knn_graph(pos, 2, 2)
    assert ei[1].tolist() == [0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7]
    assert ei[0].tolist() == [1, 2, 0, 2, 1, 0, 2, 1, 5, 6, 4, 6, 7, 5, 6, 5]
    assert ea[:8, 0].tolist() == pytest.approx([1 / 6, 3 / 6, 1 / 6, 2 / 6, 2 / 6, 3 / 6, 4 / 6, 6 / 6])
    assert float(ea[8:].max()) == 1.0


@pytest.mark.gpu
@pytest.mark.parametrize("clouds,points,k", [(1, 2, 1), (1, 3, 2), (3, 50, 7), (4, 300, 20), (32, 1024, 20), (2, 1500, 32),
                                             (1, 8192, 24)])
def test_knn_kernel_is_bit_exact(clouds, points, k):
    from mpnn_amd.knn import knn_graph
    g = torch.Generator().manual_seed(points)
    pos = torch.rand(clouds * points, 3, generator=g) * 2 - 1
    if points >= 50:
        pos[1] = pos[0]              # coincident points: zero distance, tie on the next neighbour
        pos[7] = pos[5]
        pos[points // 2 + 1] = pos[3]    # ties between points that different waves of a block scan
        pos[points - 1] = pos[3]
    want_ei, want_ea = ro.knn_graph(pos, clouds, k)
    ei, ea = knn_graph(pos.cuda(), clouds, k)
    assert torch.equal(ei.cpu(), want_ei)
    assert torch.equal(ea.cpu(), want_ea)


@pytest.mark.gpu
def test_knn_argument_checks():
    from mpnn_amd._lib import MpnnError
    from mpnn_amd.knn import knn_graph
    with pytest.raises(MpnnError):
        knn_graph(torch.zeros(10, 3, device="cuda"), 1, 10)      # k must be below the cloud size
    with pytest.raises(ValueError):
        knn_graph(torch.zeros(10, 3, device="cuda"), 3, 2)
