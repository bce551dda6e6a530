"""ovmono3d_amd - MI355X-native OVMono3D-LIFT inference path (drop-in for nightgoodl/ovmono3d's
demo.py / train_net.py --eval-only / ROIHeads3D(GDINO) plugin surface). See DESIGN.md."""
__version__ = "0.1.0"
