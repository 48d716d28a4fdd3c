// CoreSLAM.HoleMap on the GPU: the public surface of the reference class (CoreSLAM/HoleMap.cs:10-56) with the pixels
// living in device memory.  `Pixels` stays a managed ushort[] because callers read it directly
// (Simulation/MainWindow.xaml.cs:229): it is a MIRROR, refreshed by Download() (everything) or Mirror() (the rectangle
// the scans since the last refresh touched) -- CoreSLAMProcessor.Update calls Mirror() after every scan when MirrorMaps is
// set (the default, for source compatibility).
using System;
using SlamHip;

namespace CoreSLAM
{
    public class HoleMap
    {
        private readonly Handle cs;                                     // slamhip_cs of the owning processor

        /// <summary>Host mirror of the device pixels, row-major [y * Size + x] (HoleMap.cs:27).</summary>
        public readonly ushort[] Pixels;

        /// <summary>Side length in pixels (HoleMap.cs:32).</summary>
        public int Size { get; }

        /// <summary>Pixels per metre (HoleMap.cs:37): sizePixels / sizeMeters.</summary>
        public float Scale { get; }

        internal HoleMap(Handle cs, int sizePixels, float scale)
        {
            this.cs = cs;
            Size = sizePixels;
            Scale = scale;
            // (on the pinned object heap: the address never changes, so the library page-locks the array once for its partial
            // mirror copies -- slamhip_cs_holemap_mirror)
            Pixels = GC.AllocateArray<ushort>(sizePixels * sizePixels, pinned: true);
        }

        /// <summary>Refresh Pixels from the device (2 bytes per pixel over PCIe: 8 MiB at 2048 x 2048).</summary>
        public unsafe void Download()
        {
            fixed (ushort* p = Pixels)
                Native.Check(Native.slamhip_cs_holemap_download(cs.Ptr, p, (nuint)Pixels.Length));
        }

        /// <summary>Bring Pixels up to date by copying only what the updates since the last Mirror()/Download() call can have
        /// changed: the bounding rectangle of the scans drawn (the update kernel keeps it on the device).  The whole map on the
        /// first call and after Reset/Upload.  This is what CoreSLAMProcessor.Update calls when MirrorMaps is set.</summary>
        public unsafe void Mirror()
        {
            int* rect = stackalloc int[4];
            fixed (ushort* p = Pixels)
                Native.Check(Native.slamhip_cs_holemap_mirror(cs.Ptr, p, (nuint)Pixels.Length, rect));
        }

        /// <summary>Replace the device pixels with Pixels (restoring a saved map).</summary>
        public unsafe void Upload()
        {
            fixed (ushort* p = Pixels)
                Native.Check(Native.slamhip_cs_holemap_upload(cs.Ptr, p, (nuint)Pixels.Length));
        }

        /// <summary>Two pixels per byte, the 4 most significant bits of each (HoleMap.cs:44-55) -- packed on the device, so only
        /// a quarter of the map crosses PCIe.</summary>
        public unsafe byte[] GetPackedPixels()
        {
            byte[] packed = new byte[Pixels.Length / 2];
            fixed (byte* p = packed)
                Native.Check(Native.slamhip_cs_holemap_download_packed(cs.Ptr, p, (nuint)packed.Length));
            return packed;
        }
    }
}
