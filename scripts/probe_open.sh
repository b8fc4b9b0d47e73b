echo "--- import torch"; python -c "import torch" 2>&1 | grep -c amdgpu.ids
echo "--- import torch + set_num_threads"; python -c "import torch; torch.set_num_threads(1)" 2>&1 | grep -c amdgpu.ids
echo "--- oracle torch import"; python -c "import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests'); import torch; from oracle import tsadar_oracle_torch" 2>&1 | grep -c amdgpu.ids
echo "--- tensor op"; python -c "import torch; a=torch.ones(3,dtype=torch.float64); (a*2).sum().item()" 2>&1 | grep -c amdgpu.ids
echo "--- autograd"; python -c "import torch; a=torch.ones(3,dtype=torch.float64,requires_grad=True); (a*2).sum().backward()" 2>&1 | grep -c amdgpu.ids
echo "--- autograd with HIP_VISIBLE_DEVICES empty"; HIP_VISIBLE_DEVICES= python -c "import torch; a=torch.ones(3,dtype=torch.float64,requires_grad=True); (a*2).sum().backward()" 2>&1 | grep -c amdgpu.ids
echo "--- autograd with ROCR_VISIBLE_DEVICES empty"; ROCR_VISIBLE_DEVICES= python -c "import torch; a=torch.ones(3,dtype=torch.float64,requires_grad=True); (a*2).sum().backward()" 2>&1 | grep -c amdgpu.ids
true
